// Bucket sort of the (scalar, window) digit entries of an MSM.  The hot operation of an MSM is the bucket accumulation;
// putting entries in bucket order is plain data movement, so it uses rocPRIM's device radix sort (kept in its own
// translation unit: the rocPRIM headers dominate its compile time) on 64-bit keys
//      key = global bucket id << 32 | base index << 1 | negate
// over the key bits [32, 32 + ceil(log2(buckets + 1))).  The sorted keys ARE the entry list the accumulation kernel
// reads (uint2: .x = index|sign, .y = bucket).  The sort is stable, so the order inside a bucket — and therefore every
// intermediate bucket sum — is reproducible run to run (no atomics anywhere in the MSM).
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

namespace zk {

// sort 64-bit keys by their bits [32, 32 + key_bits)
void radix_sort_hi32(zkg16_ctx *ctx, const uint64_t *in, uint64_t *out, size_t count, unsigned key_bits, DevBuf &temp, const char *timer_name) {
    if (count == 0) return;
    size_t temp_bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, temp_bytes, in, out, count, 32u, 32u + key_bits, ctx->stream);
    if (e != hipSuccess) throw HipError{e, "rocprim::radix_sort_keys(size query)", __FILE__, __LINE__};
    temp.ensure(temp_bytes);
    ScopedKernelTimer kt(ctx, timer_name, (double)count, ctx->stream);
    e = rocprim::radix_sort_keys(temp.p, temp_bytes, in, out, count, 32u, 32u + key_bits, ctx->stream);
    if (e != hipSuccess) throw HipError{e, "rocprim::radix_sort_keys", __FILE__, __LINE__};
}

void msm_sort_keys(zkg16_ctx *ctx, MsmWorkspace &ws, size_t count, unsigned key_bits) {
    radix_sort_hi32(ctx, ws.keys.as<uint64_t>(), ws.entries.as<uint64_t>(), count, key_bits, ws.sort_temp, "msm_radix_sort");
}

}  // namespace zk

// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 (profiles/README.md "counter calibration").
// Four kernels whose HBM bytes are known by construction, each over a buffer far larger than L2 + Infinity Cache (2 GiB):
//   calib_stream_read16   every lane reads 16 B (global_load_dwordx4), fully coalesced: bytes = N
//   calib_stream_read4    every lane reads 4 B, fully coalesced:                        bytes = N
//   calib_gather4         every lane reads 4 B at a hashed address (one 4 B word per distinct 128 B line, no reuse):
//                         algorithmic bytes = 4 * lanes, line bytes = 64 or 128 * lanes
//   calib_stream_write16  every lane writes 16 B:                                       bytes = N
// usage: rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/_build/pmc_calib   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void calib_stream_read16(const uint4* __restrict__ in, uint32_t* __restrict__ out, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;      // never true for the fill pattern: keeps the loads alive without a write stream
}

__global__ void calib_stream_read4(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n4; i += stride) acc ^= in[i];
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void calib_gather4(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t lines, uint32_t per_lane)
{
    // lane k of the launch touches line perm(k): an odd multiplier modulo a power of two is a bijection, so no line is read twice
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < per_lane; ++r)
    {
        size_t k = t + (size_t)r * total;
        size_t line = (k * 2654435761ull) & (lines - 1);
        acc ^= in[line * 32 + (k & 31)];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// lanes 2j and 2j+1 of one load instruction read the two 64 B halves of ONE 128 B line (4 B each): one 128 B request (tallied as
// 64 B => 32 B per lane) or two 64 B requests (=> 64 B per lane)?  Tells what a sparse gather's FETCH_SIZE has to be multiplied by.
__global__ void calib_gather4_pair(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t lines, uint32_t per_lane)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < per_lane; ++r)
    {
        size_t k = (t >> 1) + (size_t)r * (total >> 1);
        size_t line = (k * 2654435761ull) & (lines - 1);
        acc ^= in[line * 32 + (t & 1) * 16 + (k & 15)];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void calib_stream_write16(uint4* __restrict__ out, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    void *buf = nullptr, *res = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&res, 256));
    CK(hipMemset(buf, 0x5a, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 32, block = 256;
    const size_t lines = bytes / 128;                      // 16 Mi lines of 128 B
    const uint32_t per_lane = 4;                           // grid * block * 4 = 8 Mi lanes-reads < lines: every read is a new line
    struct { const char* name; double bytes_known; } rows[5] = {
        {"calib_stream_read16", (double)bytes}, {"calib_stream_read4", (double)bytes},
        {"calib_gather4", (double)grid * block * per_lane * 4.0}, {"calib_stream_write16", (double)bytes},
        {"calib_gather4_pair", (double)grid * block * per_lane * 4.0}};
    for (int rep = 0; rep < 3; ++rep)
        for (int k = 0; k < 5; ++k)
        {
            CK(hipEventRecord(e0));
            if (k == 0) calib_stream_read16<<<grid, block>>>((const uint4*)buf, (uint32_t*)res, bytes / 16);
            if (k == 1) calib_stream_read4<<<grid, block>>>((const uint32_t*)buf, (uint32_t*)res, bytes / 4);
            if (k == 2) calib_gather4<<<grid, block>>>((const uint32_t*)buf, (uint32_t*)res, lines, per_lane);
            if (k == 3) calib_stream_write16<<<grid, block>>>((uint4*)buf, bytes / 16);
            if (k == 4) calib_gather4_pair<<<grid, block>>>((const uint32_t*)buf, (uint32_t*)res, lines, per_lane);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2)
                printf("{\"kernel\": \"%s\", \"known_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", rows[k].name, rows[k].bytes_known, ms, rows[k].bytes_known / ms * 1e-6);
        }
    return 0;
}

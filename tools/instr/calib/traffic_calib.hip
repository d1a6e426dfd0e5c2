// Known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (tools/instr/calib/README in
// traffic_calib.py).  Every kernel touches a table far larger than L2 + Infinity Cache exactly once per launch, in the
// access SHAPES the blend kernels use: wide coalesced reads (the calibrated baseline), gathers of 48-byte records at
// 16-byte alignment (MgrGRec), coalesced 16-byte-per-lane rows (checkpoints, pixel state), scattered stores of 48-byte
// records (three 16-byte stores per lane to a lane-private slot) and scattered 4-byte stores (the tags).
// Standalone: not part of libmanus_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

// order of the gathers / scatters: a multiplicative hash of the element index (a bijection on [0, n) for n = 2^k)
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask) { return (i * 2654435761u) & mask; }

extern "C" __global__ void cal_stream_read(const float4* __restrict__ src, float4* __restrict__ sink, size_t n16) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}

extern "C" __global__ void cal_stream_write(float4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

// n records of 48 bytes gathered from a table of (mask + 1) records, each exactly once, in hashed order: lane = one record
// (three 16-byte loads, like the blend kernels' record fetch)
extern "C" __global__ void cal_gather48(const float4* __restrict__ table, float4* __restrict__ sink, uint32_t n, uint32_t mask) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4* r = table + (size_t)perm(i, mask) * 3;
        const float4 a = r[0], b = r[1], c = r[2];
        acc.x += a.x + b.y + c.z; acc.y += a.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}

// the same with a 4-byte index list in front (sorted_gid): index coalesced, record gathered
extern "C" __global__ void cal_gather48_indexed(const float4* __restrict__ table, const uint32_t* __restrict__ idx, float4* __restrict__ sink,
                                                uint32_t n) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4* r = table + (size_t)idx[i] * 3;
        const float4 a = r[0], b = r[1], c = r[2];
        acc.x += a.x + b.y + c.z; acc.y += a.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}

// rows of 64 lanes x 16 bytes (1 KB contiguous) at hashed row positions: checkpoints / per-pixel state of a quadrant
extern "C" __global__ void cal_rows16(const float4* __restrict__ table, float4* __restrict__ sink, uint32_t nrows, uint32_t mask) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave; r < nrows; r += nw) {
        const float4 v = table[(size_t)perm(r, mask) * 64 + lane];
        acc.x += v.x; acc.y += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}

// n records of 48 bytes stored to hashed slots (three 16-byte stores per lane), optionally followed by a 4-byte tag store to a
// hashed slot of a second table
extern "C" __global__ void cal_scatter48(float4* __restrict__ table, uint32_t* __restrict__ tags, uint32_t n, uint32_t mask, int with_tag) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = perm(i, mask);
        float4* r = table + (size_t)s * 3;
        r[0] = make_float4(1.f, 2.f, 3.f, (float)i);
        r[1] = make_float4(4.f, 5.f, 6.f, 7.f);
        r[2] = make_float4(8.f, 0.f, 0.f, 0.f);
        if (with_tag) tags[s] = i;
    }
}

extern "C" __global__ void cal_scatter4(uint32_t* __restrict__ tags, uint32_t n, uint32_t mask) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) tags[perm(i, mask)] = i;
}

#define LAUNCH(k, ...) hipLaunchKernelGGL(k, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); return (int)hipGetLastError()
extern "C" int run_stream_read(const void* src, void* sink, size_t bytes, void* stream) { LAUNCH(cal_stream_read, (const float4*)src, (float4*)sink, bytes / 16); }
extern "C" int run_stream_write(void* dst, size_t bytes, void* stream) { LAUNCH(cal_stream_write, (float4*)dst, bytes / 16); }
extern "C" int run_gather48(const void* table, void* sink, uint32_t n, uint32_t mask, void* stream) { LAUNCH(cal_gather48, (const float4*)table, (float4*)sink, n, mask); }
extern "C" int run_gather48_indexed(const void* table, const void* idx, void* sink, uint32_t n, void* stream) { LAUNCH(cal_gather48_indexed, (const float4*)table, (const uint32_t*)idx, (float4*)sink, n); }
extern "C" int run_rows16(const void* table, void* sink, uint32_t nrows, uint32_t mask, void* stream) { LAUNCH(cal_rows16, (const float4*)table, (float4*)sink, nrows, mask); }
extern "C" int run_scatter48(void* table, void* tags, uint32_t n, uint32_t mask, int with_tag, void* stream) { LAUNCH(cal_scatter48, (float4*)table, (uint32_t*)tags, n, mask, with_tag); }
extern "C" int run_scatter4(void* tags, uint32_t n, uint32_t mask, void* stream) { LAUNCH(cal_scatter4, (uint32_t*)tags, n, mask); }

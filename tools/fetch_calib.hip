// tools/fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE counter on gfx950 for TWO access patterns over the same
// N bytes (round-4 review, weak #12: config 5's strip kernel reads 64-byte pieces per lane; is its raw FETCH_SIZE of
// 15.6 GB for 10 GB of bases 1.56 x or 3.1 x the data?).
//   copy_coalesced : lane l of a wave reads 16 bytes at wave_base + 16 l (one 1 KB request per wave instruction) -- the
//                    pattern the guide's "x 2" correction was derived for
//   copy_strip     : lane l reads 64 bytes (4 x 16) at lane_base = (global lane) * L + step * 64 -- every lane walks its OWN
//                    strip of L bytes, 64 bytes per loop step, like nthash_strip_kernel's base loads
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o fc --output-format csv -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void copy_coalesced(const uint4 *in, uint64_t n16, unsigned long long *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n16; i += stride) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

__global__ void copy_strip(const uint4 *in, uint64_t nbytes, uint32_t L, unsigned long long *sink) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t base = lane * L;
    if (base + L > nbytes) return;
    unsigned acc = 0;
    for (uint32_t s = 0; s < L; s += 64) {
        const uint4 *p = in + (base + s) / 16;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

int main() {
    const uint64_t N = 4ull << 30;  // 4 GiB
    uint4 *in = nullptr;
    unsigned long long *sink = nullptr;
    if (hipMalloc(&in, N) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
    (void)hipMemset(in, 1, N);
    (void)hipMemset(sink, 0, 8);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(copy_coalesced, dim3(256 * 16), dim3(256), 0, 0, in, N / 16, sink);
        for (uint32_t L : {256u, 1024u}) {
            const uint64_t lanes = N / L;
            hipLaunchKernelGGL(copy_strip, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, 0, in, N, L, sink);
        }
    }
    (void)hipDeviceSynchronize();
    printf("bytes per kernel launch: %llu\n", (unsigned long long)N);
    return 0;
}

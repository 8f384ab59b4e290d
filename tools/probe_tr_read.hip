// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): which 16-bit elements does lane l receive, given per-lane addresses?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_tr_read.hip -o tools/_probe/probe_tr && tools/_probe/probe_tr
// LDS holds element index i at element i (u16).  Pattern A: lane l reads at byte address 8*l (lanes contiguous).  Pattern B: lane l reads
// at byte address 40*l (lanes apart: shows that the exchange is between LANES of a 16-lane group, independent of the addresses).
// Output per lane: the four u16 it received.  Measurement aid for the bf16 weight-gradient kernel's operand layout; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint16_t* out, int stride_bytes) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
    const unsigned addr = base + threadIdx.x * stride_bytes;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * sizeof(uint16_t));
    const int strides[3] = {8, 40, 32};
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, strides[p]);
        uint16_t h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern stride %d bytes per lane (lane: elements received; element e lives at byte 2e)\n", strides[p]);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    hipFree(d);
    return 0;
}

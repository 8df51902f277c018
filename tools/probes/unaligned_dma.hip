// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/unaligned_dma.hip -o tools/probes/unaligned_dma ; run it on the GPU box.
// Does `buffer_load_dwordx4 ... lds` accept a global address that is only 4-byte aligned (gfx950)?  Each lane loads 16 bytes from
// base + lane * 16 + shift (shift = 0, 4, 8, 12) into LDS and the kernel copies LDS back out; the host checks the values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* dst, int shift_floats) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 4096, 0x00020000);
    const unsigned voff = (unsigned)(lane * 16 + shift_floats * 4);
    const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(m0v) : "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int sh = 0; sh < 4; ++sh) {
        hipMemset(o, 0, 1024);
        probe<<<1, 64>>>(d, o, sh);
        std::vector<float> r(256);
        hipError_t e = hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + sh);
        printf("shift %d floats: err=%d mismatches=%d first=%g %g %g %g %g\n", sh, (int)e, bad, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}

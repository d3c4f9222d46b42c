// Probe of ds_read_b64_tr_b16 (gfx950): hipcc --offload-arch=gfx950 -O2 tools/probe_tr_read.hip -o tools/tr_probe.bin; the output
// (gpurun_out/tr_probe.txt) fixed the fragment addressing of csrc/conv_wgrad.hip: in a 16-lane group, lane 4r+q supplies the
// address of 4 contiguous 16-bit elements (row r, columns 4q..4q+3); lane l receives column l of rows 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// probe ds_read_b64_tr_b16: LDS holds u16 value = its own element index; lane l reads at byte address addr[l]
__global__ void probe(const int* addr, uint16_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)lds + addr[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (uint16_t)(v >> (16 * e));
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int *d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) h_addr[l] = l * 8;                      // lane l -> elements 4l..4l+3 (contiguous rows of 4)
            if (mode == 1) h_addr[l] = (l & 15) * 128 + (l >> 4) * 8;   // row = l&15 (pitch 64 elems), col group = l>>4
            if (mode == 2) h_addr[l] = (l & 3) * 128 + (l >> 2) * 8;    // row = l&3, col group = l>>2
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr-elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
        }
    }
    return 0;
}

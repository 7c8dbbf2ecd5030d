// Developer probe (gfx950): which E8M0 scale does v_mfma_scale_f32_32x32x64_f8f6f4 apply to the byte at (lane half h, byte p)
// of an fp8 A operand, and which byte of the scale VGPR does op_sel pick?   hipcc --offload-arch=gfx950 -O2 tools/mx_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int OPSEL>
__global__ void probe(float* out, const int* a_words, const int* scales) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) {
        a[r] = a_words[lane * 8 + r];
        b[r] = 0x38383838;                                               // 1.0 everywhere
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, scales[lane], 0, 0x7F7F7F7F);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int main() {
    float* d;
    int *da, *ds;
    hipMalloc(&d, 64 * 16 * 4);
    hipMalloc(&da, 64 * 8 * 4);
    hipMalloc(&ds, 64 * 4);
    int ha[64 * 8], hs[64];
    float ho[64 * 16];
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: unit scales (sanity: every position must give 1.0); mode 1: half 0 -> bytes {2^0,2^4,2^8,2^12}, half 1 -> {2^1,2^5,2^9,2^13}
        for (int l = 0; l < 64; ++l)
            hs[l] = mode == 0 ? 0x7F7F7F7F : ((l >> 5) == 0 ? (127 | (131 << 8) | (135 << 16) | (139 << 24)) : (128 | (132 << 8) | (136 << 16) | (140 << 24)));
        hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
        for (int op = 0; op < (mode == 0 ? 1 : 4); ++op) {
            printf("mode %d opsel %d: C[row0][col0] for a single 1.0 at (half, byte):\n", mode, op);
            for (int h = 0; h < 2; ++h) {
                printf("  half %d:", h);
                for (int p = 0; p < 32; ++p) {
                    for (int i = 0; i < 64 * 8; ++i) ha[i] = 0;
                    for (int l = 0; l < 64; ++l)
                        if ((l >> 5) == h) ha[l * 8 + (p >> 2)] = 0x38 << (8 * (p & 3));
                    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
                    switch (op) {
                        case 0: probe<0><<<1, 64>>>(d, da, ds); break;
                        case 1: probe<1><<<1, 64>>>(d, da, ds); break;
                        case 2: probe<2><<<1, 64>>>(d, da, ds); break;
                        default: probe<3><<<1, 64>>>(d, da, ds); break;
                    }
                    hipError_t err = hipDeviceSynchronize();
                    if (err != hipSuccess) { printf("error %s\n", hipGetErrorString(err)); return 1; }
                    hipMemcpy(ho, d, sizeof(ho), hipMemcpyDeviceToHost);
                    printf(" %g", ho[0]);
                }
                printf("\n");
            }
        }
    }
    return 0;
}

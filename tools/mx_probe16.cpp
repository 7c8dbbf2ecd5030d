// Developer probe (gfx950): block scales on the B side of v_mfma_scale_f32_16x16x128_f8f6f4 -- which lane's scale byte (and which byte, by
// op_sel) multiplies the 32 operand bytes lane (n = l % 16, k-group g = l / 16) holds.   hipcc --offload-arch=gfx950 -O2 tools/mx_probe16.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int OPSEL>
__global__ void probe(float* out, const int* b_words, const int* scales) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) {
        b[r] = b_words[lane * 8 + r];
        a[r] = 0x38383838;                                               // e4m3 1.0 everywhere
    }
    f32x4 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, OPSEL, scales[lane]);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
    float* d;
    int *db, *ds;
    hipMalloc(&d, 64 * 4 * 4);
    hipMalloc(&db, 64 * 8 * 4);
    hipMalloc(&ds, 64 * 4);
    int hb[64 * 8], hs[64];
    float ho[64 * 4];
    // lane l = (n = l % 16, g = l / 16): scale bytes {2^(4g), 2^(4g+1), 2^(4g+2), 2^(4g+3)}; column n = 5 gets an extra factor 2^16 to tell columns apart
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, e = 127 + 4 * g + ((l & 15) == 5 ? 16 : 0);
        hs[l] = e | ((e + 1) << 8) | ((e + 2) << 16) | ((e + 3) << 24);
    }
    hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
    for (int op = 0; op < 4; ++op) {
        printf("opsel %d: log2 C[row 0][col 0] / C[row 0][col 5] for a single 1.0 in B at (k-group g, byte p), all columns:\n", op);
        for (int g = 0; g < 4; ++g) {
            printf("  g %d:", g);
            for (int p = 0; p < 32; p += 5) {
                for (int i = 0; i < 64 * 8; ++i) hb[i] = 0;
                for (int l = 0; l < 64; ++l)
                    if ((l >> 4) == g) hb[l * 8 + (p >> 2)] = 0x38 << (8 * (p & 3));
                hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
                switch (op) {
                    case 0: probe<0><<<1, 64>>>(d, db, ds); break;
                    case 1: probe<1><<<1, 64>>>(d, db, ds); break;
                    case 2: probe<2><<<1, 64>>>(d, db, ds); break;
                    default: probe<3><<<1, 64>>>(d, db, ds); break;
                }
                if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
                hipMemcpy(ho, d, sizeof(ho), hipMemcpyDeviceToHost);
                // C layout 16x16: lane l holds col l % 16, rows 4 * (l / 16) + r
                printf("  p%-2d %5.1f/%5.1f", p, __builtin_log2f(ho[0 * 4 + 0]), __builtin_log2f(ho[5 * 4 + 0]));
            }
            printf("\n");
        }
    }
    return 0;
}

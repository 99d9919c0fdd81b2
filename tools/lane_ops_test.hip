// Which cheap cross-lane instruction realises "value of lane (l xor m)" on gfx950?  Checked against ds_bpermute.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ int ref_xor(int x, int m) { return __builtin_amdgcn_ds_bpermute(((int)threadIdx.x ^ m) << 2, x); }
__global__ void k(int* out) {
  const int l = threadIdx.x, x = 1000 + l;
  int* o = out + l * 16;
  auto s32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  o[0] = ref_xor(x, 32); o[1] = s32[0]; o[2] = s32[1];
  o[3] = ref_xor(x, 16); o[4] = s16[0]; o[5] = s16[1];
  o[6] = ref_xor(x, 8);  o[7] = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);
  int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
  t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);
  o[8] = ref_xor(x, 4);  o[9] = t;
  o[10] = ref_xor(x, 2); o[11] = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);
  o[12] = ref_xor(x, 1); o[13] = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);
}
int main() {
  int* d; hipMalloc(&d, 64 * 16 * 4);
  k<<<1, 64>>>(d);
  int h[64 * 16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  auto col = [&](int c, int l) { return h[l * 16 + c]; };
  for (int m : {32, 16}) {
    const int base = m == 32 ? 0 : 3;
    int ok0 = 0, ok1 = 0, lo0 = 0, lo1 = 0;
    for (int l = 0; l < 64; ++l) { ok0 += col(base + 1, l) == col(base, l); ok1 += col(base + 2, l) == col(base, l); }
    printf("xor %d: result[0] matches in %d lanes, result[1] in %d lanes; lanes where [0] matches:", m, ok0, ok1);
    for (int l = 0; l < 64; ++l) if (col(base + 1, l) == col(base, l)) printf(" %d", l);
    printf("\n");
  }
  const int pairs[4][3] = {{8, 6, 7}, {4, 8, 9}, {2, 10, 11}, {1, 12, 13}};
  for (auto& p : pairs) { int ok = 0; for (int l = 0; l < 64; ++l) ok += col(p[1], l) == col(p[2], l); printf("xor %d via DPP: %d / 64 lanes match\n", p[0], ok); }
  return 0;
}

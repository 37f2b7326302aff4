// scratch: occupancy + block-count scaling of the conv kernel (not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "yv4.h"
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d, LDS/block max %zu, LDS/CU %zu, clock %d kHz\n", p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.clockRate);
  // time conv with M chosen so that tiles = k*256 for tile 128x128 (Cout=128 -> 1 n-tile)
  int Cin = 256, Cout = 128, H = 32;  // M = N*32*32 = N*1024 -> tiles_m = N*8
  for (int tile : {8, 9, 10, 11, 12, 13}) {
    for (int N : {64, 192}) {
      size_t xs = (size_t)N * H * H * Cin, ws = (size_t)Cout * 9 * Cin, ys = (size_t)N * H * H * Cout;
      float *x, *w, *y, *s, *t;
      hipMalloc(&x, xs * 4); hipMalloc(&w, ws * 4); hipMalloc(&y, ys * 4); hipMalloc(&s, Cout * 4); hipMalloc(&t, Cout * 4);
      hipMemset(x, 0, xs * 4); hipMemset(w, 0, ws * 4); hipMemset(s, 0, Cout * 4); hipMemset(t, 0, Cout * 4);
      yv4_conv_desc d = {};
      d.N = N; d.H = H; d.W = H; d.Cin = Cin; d.Ho = H; d.Wo = H; d.Cout = Cout; d.KH = d.KW = 3; d.stride = 1; d.pad = 1;
      d.x_cstride = Cin; d.y_cstride = Cout; d.act1 = 1; d.tile = tile;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, 0);
        int rc = yv4_conv_bn_act_fwd(&d, x, w, s, t, nullptr, nullptr, nullptr, y, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("err %s\n", yv4_last_error()); return 1; }
        if (r && ms < best) best = ms;
      }
      int bm = 64, bn = 64;
      long tiles = ((long)N * H * H / bm) * (Cout / bn);
      double fl = 2.0 * N * H * H * Cout * 9.0 * Cin;
      printf("tile %d N %3d tiles %5ld (%.2f per CU): %8.1f us  %6.1f TF\n", tile, N, tiles, tiles / 256.0, best * 1e3, fl / best / 1e9);
      hipFree(x); hipFree(w); hipFree(y); hipFree(s); hipFree(t);
    }
  }
  return 0;
}

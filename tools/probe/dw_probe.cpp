// Grouped weight-gradient launch of one encoder layer in isolation (kernel experiments on k_dw_grouped): loads a (variant) library,
// runs the five-job launch at a bench-like shape and prints the time per launch and a checksum of the partial tiles.
//   hipcc -O2 -o dw_probe dw_probe.cpp -ldl ;  ./dw_probe <lib.so> [rows] [d] [S]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../gd-mae_amd/csrc/dw_grouped.h"
typedef int (*fn_t)(hipStream_t, GdDwGroup&, long long, long long, int);
static unsigned short bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
int main(int argc, char** argv) {
  const char* lib = argv[1];
  const long long rows = argc > 2 ? atoll(argv[2]) : 43008;
  const int d = argc > 3 ? atoi(argv[3]) : 256, S = argc > 4 ? atoi(argv[4]) : 0, ff = 2 * d;
  void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  fn_t fn = (fn_t)dlsym(h, "_Z15gd_dw_grouped_sP12ihipStream_tR9GdDwGroupxxi");
  if (!fn) { printf("no symbol\n"); return 1; }
  const int MN[5][2] = {{d, ff}, {ff, d}, {d, d}, {2 * d, d}, {d, d}};
  GdDwGroup A{};
  A.n_jobs = 5;
  std::vector<float*> parts;
  size_t part_elems = 0;
  for (int j = 0; j < 5; ++j) {
    const int M = MN[j][0], N = MN[j][1];
    std::vector<unsigned short> hg((size_t)rows * M), hx((size_t)rows * N);
    unsigned s = 12345u + j;
    for (auto& v : hg) { s = s * 1664525u + 1013904223u; v = bf(((int)(s >> 9) % 2001 - 1000) * 1e-3f); }
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = bf(((int)(s >> 9) % 2001 - 1000) * 1e-3f); }
    void *G, *X; float *P, *C;
    hipMalloc(&G, hg.size() * 2); hipMalloc(&X, hx.size() * 2);
    hipMemcpy(G, hg.data(), hg.size() * 2, hipMemcpyHostToDevice); hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&P, (size_t)64 * M * N * 4); hipMalloc(&C, (size_t)64 * M * 4);
    A.job[j] = GdDwJob{G, X, M, N, P, (j == 1 || j >= 3) ? C : nullptr, 0, nullptr, 0, 0, 0, 0};
    parts.push_back(P);
    part_elems += (size_t)M * N;
  }
  hipStream_t st; hipStreamCreate(&st);
  for (int i = 0; i < 3; ++i) if (fn(st, A, rows, rows, S)) { printf("launch failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 30;
  hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) fn(st, A, rows, rows, S);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // checksum: sum over slices of job 0's partial tiles
  std::vector<float> hp((size_t)A.S * MN[0][0] * MN[0][1]);
  hipMemcpy(hp.data(), parts[0], hp.size() * 4, hipMemcpyDeviceToHost);
  double cs = 0; for (size_t i = 0; i < hp.size(); ++i) cs += hp[i] * (double)((i % 97) + 1);
  const double fl = 2.0 * rows * part_elems;
  printf("%s rows %lld d %d S %d wgs %d: %.1f us/launch  %.0f TFLOP/s  checksum %.6e\n", lib, rows, d, A.S, A.tiles_total * A.S, ms / n * 1e3, fl / (ms / n * 1e-3) / 1e12, cs);
  return 0;
}

// Probe (round 4): phase stamps of bins_cells_kernel (gp_binning.hip) -- the GP_SORT_TRACE build of the binning, on 1 M / 2 M uniform random points.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGP_SORT_TRACE -I gtsam_points_amd/csrc -I include -o /tmp/bins_probe scripts/probe/bins_probe.hip -ldl && /tmp/bins_probe
#include "gp_runtime.hip"
#include "gp_binning.hip"

extern "C" int gp_source_mirror_invalidate(const void*) { return 0; }  // (gp_cloud.hip is not part of the probe)

#include <algorithm>
#include <random>

static double med(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int run(int n, double cell) {
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
  std::vector<float> h(3 * (size_t)n);
  std::mt19937 rng(1);
  // a ground sheet of 136 m x 136 m, a few centimetres thick: ~74 k voxels at 0.5 m like the bench map (27 points per voxel at 2 M points)
  std::uniform_real_distribution<float> ux(-68.f, 68.f), uz(-0.1f, 0.1f);
  for (int i = 0; i < n; i++) h[3 * i] = ux(rng), h[3 * i + 1] = ux(rng), h[3 * i + 2] = uz(rng);
  float* d;
  if (hipMalloc(&d, 12 * (size_t)n) != hipSuccess) return 1;
  (void)hipMemcpy(d, h.data(), 12 * (size_t)n, hipMemcpyHostToDevice);
  const int tiles = (n + 4095) / 4096;
  unsigned long long* trace;
  (void)hipMalloc(&trace, 8 * 16 * (size_t)tiles);
  std::vector<unsigned long long> t(16 * (size_t)tiles);
  for (int rep = 0; rep < 4; rep++) {
    gp::PointBins bins;
    bool too_large = false;
    // the sort's kernels stamp the same buffer: only the last kernel's (the cells kernel's) stamps survive in slots 0..4
    (void)hipMemset(trace, 0, 8 * 16 * (size_t)tiles);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gp::g_sort_trace), &trace, sizeof(trace));
    if (gp::bin_points(d, n, 1.0 / cell, s, &bins, &too_large) != 0) return 1;
    (void)hipMemcpy(t.data(), trace, 8 * t.size(), hipMemcpyDeviceToHost);
    if (rep == 3) {
      unsigned long long first = ~0ull, last = 0;
      for (int i = 0; i < tiles; i++) first = std::min(first, t[16 * (size_t)i]), last = std::max(last, t[16 * (size_t)i + 4]);
      printf("n %d cell %.2f (%d cells): cells kernel first ticket -> last tile done %.2f us; phases (median / max over tiles, us):", n, cell, bins.num_cells, (last - first) / 100.0);
      const char* names[4] = {"load+flags+scan", "publish+prefix", "ordinals", "cells"};
      for (int k = 0; k < 4; k++) {
        std::vector<double> dd;
        for (int i = 0; i < tiles; i++) dd.push_back((double)(long long)(t[16 * (size_t)i + k + 1] - t[16 * (size_t)i + k]) / 100.0);
        printf(" %s %.2f / %.2f;", names[k], med(dd), *std::max_element(dd.begin(), dd.end()));
      }
      {
        // the key kernel's own stamps (slots 8 .. 11; its tiles are the same 4096 points)
        unsigned long long kfirst = ~0ull, klast = 0;
        for (int i = 0; i < tiles; i++) kfirst = std::min(kfirst, t[16 * (size_t)i + 8]), klast = std::max(klast, t[16 * (size_t)i + 11]);
        printf(" | key kernel: first start -> last flush issued %.2f us; phases (median / max):", (klast - kfirst) / 100.0);
        const char* kn[3] = {"load+key+count", "barrier", "flush issue"};
        for (int k = 0; k < 3; k++) {
          std::vector<double> dd;
          for (int i = 0; i < tiles; i++) dd.push_back((double)(long long)(t[16 * (size_t)i + 9 + k] - t[16 * (size_t)i + 8 + k]) / 100.0);
          printf(" %s %.2f / %.2f;", kn[k], med(dd), *std::max_element(dd.begin(), dd.end()));
        }
        std::vector<double> ks;
        for (int i = 0; i < tiles; i++) ks.push_back((t[16 * (size_t)i + 8] - kfirst) / 100.0);
        printf(" start median %.2f max %.2f |", med(ks), *std::max_element(ks.begin(), ks.end()));
      }
      std::vector<double> st;
      for (int i = 0; i < tiles; i++) st.push_back((t[16 * (size_t)i] - first) / 100.0);
      printf(" ticket time median %.2f max %.2f\n", med(st), *std::max_element(st.begin(), st.end()));
    }
  }
  return 0;
}

int main() {
  if (run(2000000, 0.5)) return 1;   // voxel map: ~27 points per cell
  if (run(1000000, 0.25)) return 1;  // search grid: few points per cell
  return 0;
}

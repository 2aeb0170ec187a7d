// test_host.cpp -- end-to-end driver of the C++ mirror (runs on the GPU box; compiled by __graft_entry__.build()).
// Mirrors the shape of src/test/test_matching_cost_factors.cpp: frames -> voxel maps -> IntegratedVGICPFactorGPU through a
// StreamTempBufferRoundRobin -> LinearizationHook-driven Levenberg-Marquardt -> pose error gate.
#include <cmath>
#include <array>
#include <cstdio>
#include <cstring>
#include <random>

#include "nonlinear_factor_set_gpu.hpp"

using namespace gtsam_points;

static void make_room(int n, unsigned seed, const gtsam::Pose3& frame_from_world, std::vector<float>& pts, std::vector<float>& covs) {
  // three orthogonal walls of a room corner + floor clutter; covariance I - 0.999 n n^T from the surface normal
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::normal_distribution<double> noise(0.0, 0.01);
  pts.resize(3 * (size_t)n);
  covs.resize(9 * (size_t)n);
  for (int i = 0; i < n; i++) {
    const int wall = i % 3;
    double p[3], nrm[3] = {0, 0, 0};
    const double a = 1.0 + 9.0 * u(rng), b = 1.0 + 9.0 * u(rng);
    if (wall == 0) { p[0] = a; p[1] = b; p[2] = 0.3 * std::sin(0.8 * a) * 0 + noise(rng); nrm[2] = 1; }
    if (wall == 1) { p[0] = a; p[2] = 0.3 * b; p[1] = 10.5 + noise(rng); nrm[1] = 1; }
    if (wall == 2) { p[1] = a; p[2] = 0.3 * b; p[0] = 10.5 + noise(rng); nrm[0] = 1; }
    for (int r = 0; r < 3; r++) {
      pts[3 * (size_t)i + r] = (float)(frame_from_world.R(r, 0) * p[0] + frame_from_world.R(r, 1) * p[1] + frame_from_world.R(r, 2) * p[2] + frame_from_world.t(r));
    }
    double nf[3];
    for (int r = 0; r < 3; r++) nf[r] = frame_from_world.R(r, 0) * nrm[0] + frame_from_world.R(r, 1) * nrm[1] + frame_from_world.R(r, 2) * nrm[2];
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) covs[9 * (size_t)i + c * 3 + r] = (float)((r == c ? 1.0 : 0.0) - 0.999 * nf[r] * nf[c]);
  }
}

static bool solve6(gtsam::Matrix6 A, gtsam::Vector6 b, gtsam::Vector6& x) {  // Gaussian elimination with partial pivoting (column-major A)
  int n = 6;
  for (int k = 0; k < n; k++) {
    int piv = k;
    for (int r = k + 1; r < n; r++)
      if (std::fabs(A[k * 6 + r]) > std::fabs(A[k * 6 + piv])) piv = r;
    if (std::fabs(A[k * 6 + piv]) < 1e-12) return false;
    if (piv != k) {
      for (int c = 0; c < n; c++) std::swap(A[c * 6 + k], A[c * 6 + piv]);
      std::swap(b[k], b[piv]);
    }
    for (int r = k + 1; r < n; r++) {
      const double f = A[k * 6 + r] / A[k * 6 + k];
      for (int c = k; c < n; c++) A[c * 6 + r] -= f * A[c * 6 + k];
      b[r] -= f * b[k];
    }
  }
  for (int r = n - 1; r >= 0; r--) {
    double s = b[r];
    for (int c = r + 1; c < n; c++) s -= A[c * 6 + r] * x[c];
    x[r] = s / A[r * 6 + r];
  }
  return true;
}

#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

int main() {
  int ndev = 0;
  CHECK(gp_device_count(&ndev) == GP_OK && ndev > 0);
  const int N = 30000;
  gtsam::Pose3 world;  // identity: target frame == world
  const gtsam::Pose3 T_true = gtsam::Pose3::Expmap({0.02, -0.03, 0.05, 0.15, -0.1, 0.05});  // source sensor pose in the target frame
  std::vector<float> tp, tc, sp, sc;
  make_room(N, 1, world, tp, tc);
  make_room(N, 2, T_true.inverse(), sp, sc);  // source points expressed in the source frame

  auto target = std::make_shared<PointCloudGPU>();
  target->add_points_gpu<float, 3>(tp.data(), N);
  target->add_covs_gpu<float, 3>(tc.data(), N);
  auto source = std::make_shared<PointCloudGPU>();
  source->add_points_gpu<float, 3>(sp.data(), N);
  source->add_covs_gpu<float, 3>(sc.data(), N);

  auto voxels = std::make_shared<GaussianVoxelMapGPU>(0.5f);
  voxels->insert(*target);
  CHECK(voxels->voxelmap_info.num_voxels > 100 && voxels->buckets != nullptr && voxels->loaded_on_gpu());
  const auto means = download_voxel_means(*voxels);
  const auto buckets = download_buckets(*voxels);
  CHECK((int)means.size() == 3 * voxels->voxelmap_info.num_voxels && (int)buckets.size() == voxels->voxelmap_info.num_buckets);
  const double self_overlap = overlap_gpu(voxels, target, gtsam::Pose3().matrix().data());
  CHECK(self_overlap > 0.99);  // test_voxelmap.cpp:226

  // save / load round trip
  voxels->save_compact("/tmp/gp_host_voxels.bin");
  auto loaded = GaussianVoxelMapGPU::load("/tmp/gp_host_voxels.bin");
  CHECK(loaded && loaded->voxelmap_info.num_voxels == voxels->voxelmap_info.num_voxels);
  CHECK(std::fabs(overlap_gpu(loaded, source, T_true.matrix().data()) - overlap_gpu(voxels, source, T_true.matrix().data())) < 1e-3);

  // overload set of overlap_gpu + merge_frames_gpu (the callers that decide which submap pairs get factors)
  {
    std::array<double, 16> I16, T16;
    std::memcpy(I16.data(), gtsam::Pose3().matrix().data(), sizeof(double) * 16);
    std::memcpy(T16.data(), T_true.matrix().data(), sizeof(double) * 16);
    const std::vector<GaussianVoxelMap::ConstPtr> two{voxels, loaded};
    const double u = overlap_gpu(two, source, std::vector<std::array<double, 16>>{T16, T16});
    CHECK(std::fabs(u - overlap_gpu(voxels, source, T16.data())) < 1e-12);  // the same map twice: union == single
    const auto rates = overlap_gpu(two, std::vector<PointCloud::ConstPtr>{target, source}, std::vector<std::array<double, 16>>{I16, T16});
    CHECK(rates.size() == 2 && rates[0] == self_overlap && std::fabs(rates[1] - u) < 1e-12);
    // double 4-vector / 4x4 inputs go through the same pack kernels
    std::vector<double> p4(4 * (size_t)N), c4(16 * (size_t)N, 0.0);
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < 3; k++) p4[4 * (size_t)i + k] = sp[3 * (size_t)i + k];
      p4[4 * (size_t)i + 3] = 1.0;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) c4[16 * (size_t)i + 4 * c + r] = sc[9 * (size_t)i + 3 * c + r];
    }
    auto source4 = std::make_shared<PointCloudGPU>();
    source4->add_points_gpu<double, 4>(p4.data(), N);
    source4->add_covs_gpu<double, 4>(c4.data(), N);
    CHECK(overlap_gpu(voxels, source4, T16.data()) == overlap_gpu(voxels, source, T16.data()));
    auto merged = merge_frames_gpu({I16, T16}, std::vector<PointCloud::ConstPtr>{target, source4}, 0.25);
    CHECK(merged->size() > 1000 && merged->size() < (size_t)(2 * N) && merged->points_gpu && merged->covs_gpu);
    CHECK(overlap_gpu(voxels, merged, I16.data()) > 0.9);  // both frames land on the target's surfaces
  }

  // OffloadableGPU round trip on a cloud (the factor-level protocol is exercised further down)
  {
    const double before = overlap_gpu(voxels, source, T_true.matrix().data());
    CHECK(source->loaded_on_gpu() && source->memory_usage_gpu() == (size_t)48 * N);
    CHECK(source->offload_gpu() && !source->offload_gpu() && !source->loaded_on_gpu() && source->points_gpu == nullptr);
    CHECK(source->touch() && source->loaded_on_gpu() && !source->reload_gpu());
    CHECK(overlap_gpu(voxels, source, T_true.matrix().data()) == before);
  }

  // factor through the round-robin pool + the linearisation hook, as the applications do
  LinearizationHook::register_hook([] { return create_nonlinear_factor_set_gpu(); });
  StreamTempBufferRoundRobin roundrobin(4);
  auto sb = roundrobin.get_stream_buffer();
  gtsam::NonlinearFactorGraph graph;
  auto factor = graph.emplace_shared<IntegratedVGICPFactorGPU>(world, 1, voxels, source, sb.first, sb.second);  // unary: fixed target
  auto sb2 = roundrobin.get_stream_buffer();
  auto factor_b = graph.emplace_shared<IntegratedVGICPFactorGPU>(0, 1, voxels, source, sb2.first, sb2.second);  // binary
  LinearizationHook hook(graph);
  CHECK(hook.size() == 2);

  gtsam::Values values;
  values.insert(0, world);
  values.insert(1, T_true * gtsam::Pose3::Expmap({0.03, -0.02, 0.04, 0.1, 0.08, -0.05}));
  double lambda = 1e-5, err = 0.0;
  for (int iter = 0; iter < 30; iter++) {
    hook.linearize(values);
    auto hf = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor->linearize(values));
    auto hb = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor_b->linearize(values));
    CHECK(hf && hb && hb->binary);
    // the binary factor's source block equals the unary factor's (same delta): batch consistency
    for (int k = 0; k < 36; k++) CHECK(std::fabs(hb->G22[k] - hf->G11[k]) <= 1e-9 * (1.0 + std::fabs(hf->G11[k])));
    err = hf->f;
    bool improved = false;
    for (int t = 0; t < 10 && !improved; t++) {
      gtsam::Matrix6 A = hf->G11;
      for (int d = 0; d < 6; d++) A[d * 6 + d] *= (1.0 + lambda);
      gtsam::Vector6 dx{};
      CHECK(solve6(A, hf->g1, dx));
      gtsam::Values trial = values;
      trial.update(1, values.at<gtsam::Pose3>(1).retract(dx));
      hook.error(trial);
      const double new_err = factor->error(trial);
      (void)factor_b->error(trial);
      if (new_err < err) {
        values = trial;
        lambda = std::max(lambda / 10.0, 1e-12);
        improved = true;
        if ((err - new_err) / err < 1e-6) iter = 1000;
      } else {
        lambda *= 10.0;
      }
    }
    if (!improved) break;
  }
  const gtsam::Pose3 d = T_true.inverse() * values.at<gtsam::Pose3>(1);
  const double trace = d.R(0, 0) + d.R(1, 1) + d.R(2, 2);
  const double ang = std::acos(std::min(1.0, std::max(-1.0, (trace - 1.0) / 2.0)));
  const double trans = std::sqrt(d.t(0) * d.t(0) + d.t(1) * d.t(1) + d.t(2) * d.t(2));
  std::printf("rot err %.5f rad, trans err %.5f m, inliers %d / %d, gpu linearizations %d, evaluations %d\n", ang, trans, factor->num_inliers(), N, hook.linearization_count(),
              hook.evaluation_count());
  CHECK(ang < 0.015 && trans < 0.15);  // the reference's gate (test_matching_cost_factors.cpp:227-228)
  CHECK(factor->inlier_fraction() > 0.8 && hook.linearization_count() > 0 && hook.evaluation_count() > 0);
  auto cl = factor->clone();
  CHECK(cl->keys().size() == 1 && cl->dim() == 6);
  // offload / reload
  CHECK(voxels->offload_gpu() && !voxels->loaded_on_gpu() && voxels->buckets == nullptr);
  CHECK(voxels->reload_gpu() && voxels->loaded_on_gpu());
  // the factor-level protocol: both operands offloaded by the application, the factor brings them back and gets the same answer
  {
    hook.linearize(values);
    auto ref_lin = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor_b->linearize(values));
    (void)factor->linearize(values);
    factor->set_enable_offloading(true);
    factor_b->set_enable_offloading(true);
    CHECK(voxels->offload_gpu() && source->offload_gpu());
    hook.linearize(values);
    auto again = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor_b->linearize(values));
    (void)factor->linearize(values);
    CHECK(voxels->loaded_on_gpu() && source->loaded_on_gpu());
    for (int k = 0; k < 36; k++) CHECK(again->G11[k] == ref_lin->G11[k] && again->G22[k] == ref_lin->G22[k] && again->G12[k] == ref_lin->G12[k]);
    CHECK(again->f == ref_lin->f);
  }
  roundrobin.sync_all();
  std::printf("HOST_TEST_OK\n");
  return 0;
}

// test_host.cpp -- end-to-end driver of the C++ mirror, compiled AGAINST THE REFERENCE'S OWN HEADERS (see the Makefile) and run on
// the GPU box by tests/test_host_gpu.py.  Mirrors the shape of src/test/test_matching_cost_factors.cpp and test_voxelmap.cpp:
// frames -> voxel maps -> IntegratedVGICPFactorGPU through a StreamTempBufferRoundRobin -> the reference's LinearizationHook ->
// Levenberg-Marquardt -> pose error gate; plus the overlap_gpu overload set, merge_frames_gpu, offloading, the generic
// NonlinearFactorGPU protocol and the sharded (multi-device) factor set.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "levenberg_marquardt_graph_gpu.hpp"
#include "nonlinear_factor_set_gpu.hpp"

using namespace gtsam_points;
using Vec3f = Eigen::Vector3f;
using Mat3f = Eigen::Matrix3f;

static void make_room(int n, unsigned seed, const gtsam::Pose3& frame_from_world, std::vector<Vec3f>& pts, std::vector<Mat3f>& covs) {
  // three orthogonal walls of a room corner; covariance I - 0.999 n n^T from the surface normal
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::normal_distribution<double> noise(0.0, 0.01);
  const Eigen::Matrix4d F = frame_from_world.matrix();
  pts.resize((size_t)n);
  covs.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    const int wall = i % 3;
    double p[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    const double a = 1.0 + 9.0 * u(rng), b = 1.0 + 9.0 * u(rng);
    if (wall == 0) { p[0] = a; p[1] = b; p[2] = noise(rng); nrm[2] = 1; }
    if (wall == 1) { p[0] = a; p[2] = 0.3 * b; p[1] = 10.5 + noise(rng); nrm[1] = 1; }
    if (wall == 2) { p[1] = a; p[2] = 0.3 * b; p[0] = 10.5 + noise(rng); nrm[0] = 1; }
    double nf[3];
    for (int r = 0; r < 3; r++) {
      pts[(size_t)i][r] = (float)(F(r, 0) * p[0] + F(r, 1) * p[1] + F(r, 2) * p[2] + F(r, 3));
      nf[r] = F(r, 0) * nrm[0] + F(r, 1) * nrm[1] + F(r, 2) * nrm[2];
    }
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) covs[(size_t)i](r, c) = (float)((r == c ? 1.0 : 0.0) - 0.999 * nf[r] * nf[c]);
  }
}

static bool solve6(gtsam::Matrix6 A, gtsam::Vector6 b, gtsam::Vector6& x) {  // Gaussian elimination with partial pivoting
  const int n = 6;
  for (int k = 0; k < n; k++) {
    int piv = k;
    for (int r = k + 1; r < n; r++)
      if (std::fabs(A(r, k)) > std::fabs(A(piv, k))) piv = r;
    if (std::fabs(A(piv, k)) < 1e-12) return false;
    if (piv != k) {
      for (int c = 0; c < n; c++) std::swap(A(k, c), A(piv, c));
      std::swap(b[k], b[piv]);
    }
    for (int r = k + 1; r < n; r++) {
      const double f = A(r, k) / A(k, k);
      for (int c = k; c < n; c++) A(r, c) -= f * A(k, c);
      b[r] -= f * b[k];
    }
  }
  for (int r = n - 1; r >= 0; r--) {
    double s = b[r];
    for (int c = r + 1; c < n; c++) s -= A(r, c) * x[c];
    x[r] = s / A(r, r);
  }
  return true;
}

static gtsam::Vector6 xi6(double a, double b, double c, double d, double e, double f) {
  gtsam::Vector6 v;
  v[0] = a; v[1] = b; v[2] = c; v[3] = d; v[4] = e; v[5] = f;
  return v;
}

#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
      return 1;                                                        \
    }                                                                  \
  } while (0)

// a third-party NonlinearFactorGPU (wraps the VGICP factor so that the set cannot take its batched fast path): must keep working
// through the reference's staging-buffer protocol (cuda/nonlinear_factor_set_gpu.cpp:64-139)
class WrappedFactor : public NonlinearFactorGPU {
public:
  explicit WrappedFactor(const std::shared_ptr<IntegratedVGICPFactorGPU>& inner) : NonlinearFactorGPU(inner->keys()), inner(inner) {}
  size_t dim() const override { return 6; }
  double error(const gtsam::Values& v) const override { return inner->error(v); }
  gtsam::GaussianFactor::shared_ptr linearize(const gtsam::Values& v) const override { return inner->linearize(v); }
  size_t linearization_input_size() const override { return inner->linearization_input_size(); }
  size_t linearization_output_size() const override { return inner->linearization_output_size(); }
  size_t evaluation_input_size() const override { return inner->evaluation_input_size(); }
  size_t evaluation_output_size() const override { return inner->evaluation_output_size(); }
  void set_linearization_point(const gtsam::Values& v, void* b) override { inner->set_linearization_point(v, b); }
  void issue_linearize(const void* a, const void* b, void* c) override { inner->issue_linearize(a, b, c); }
  void store_linearized(const void* b) override { inner->store_linearized(b); }
  void set_evaluation_point(const gtsam::Values& v, void* b) override { inner->set_evaluation_point(v, b); }
  void issue_compute_error(const void* a, const void* b, const void* c, const void* d, void* e) override { inner->issue_compute_error(a, b, c, d, e); }
  void store_computed_error(const void* b) override { inner->store_computed_error(b); }
  void sync() override { inner->sync(); }
  std::shared_ptr<IntegratedVGICPFactorGPU> inner;
};

// a factor the GPU set must decline (isam2_ext.cpp:110-116: `if (hook->add(f)) ... else f->linearize(theta)`)
class HostOnlyFactor : public gtsam::NonlinearFactor {
public:
  HostOnlyFactor() : gtsam::NonlinearFactor(gtsam::KeyVector{1}) {}
  size_t dim() const override { return 6; }
  double error(const gtsam::Values&) const override { return 0.0; }
  gtsam::GaussianFactor::shared_ptr linearize(const gtsam::Values&) const override { return nullptr; }
};

int main() {
  int ndev = 0;
  CHECK(gp_device_count(&ndev) == GP_OK && ndev > 0);
  const int N = 30000;
  const gtsam::Pose3 world;  // identity: target frame == world
  const gtsam::Pose3 T_true = gtsam::Pose3::Expmap(xi6(0.02, -0.03, 0.05, 0.15, -0.1, 0.05));  // source sensor pose in the target frame
  const Eigen::Isometry3d I_iso = Eigen::Isometry3d::Identity(), T_iso(T_true.matrix());
  std::vector<Vec3f> tp, sp;
  std::vector<Mat3f> tc, sc;
  make_room(N, 1, world, tp, tc);
  make_room(N, 2, T_true.inverse(), sp, sc);  // source points expressed in the source frame

  auto target = std::make_shared<PointCloudGPU>();
  target->add_points_gpu(tp);
  target->add_covs_gpu(tc);
  auto source = std::make_shared<PointCloudGPU>();
  source->add_points_gpu(sp);
  source->add_covs_gpu(sc);
  CHECK(target->has_points_gpu() && target->check_covs_gpu());  // the reference's own PointCloud members (types/point_cloud.cpp)

  {
    // PointCloudGPU's CPU + GPU forms, clone() and download_points() (types/point_cloud_gpu.hpp:41-129, test_types.cpp's frame checks)
    PointCloudGPU both(sp);  // constructor = add_points: CPU storage (Vector4d, w = 1) and GPU storage
    CHECK(both.has_points() && both.has_points_gpu() && both.size() == sp.size());
    CHECK(both.points[7](0) == (double)sp[7](0) && both.points[7](2) == (double)sp[7](2) && both.points[7](3) == 1.0);
    both.add_covs(sc);
    CHECK(both.has_covs() && both.has_covs_gpu() && both.covs[5](1, 2) == (double)sc[5](1, 2) && both.covs[5](3, 3) == 0.0);
    std::vector<double> stamps(sp.size());
    for (size_t i = 0; i < stamps.size(); i++) stamps[i] = 1e-4 * (double)i;
    both.add_times(stamps);
    CHECK(both.has_times() && both.has_times_gpu() && both.times[10] == stamps[10]);
    std::vector<float> t_back(sp.size());
    CHECK(gp_memcpy_d2h(t_back.data(), both.times_gpu, sizeof(float) * t_back.size(), nullptr) == GP_OK && gp_stream_synchronize(nullptr) == GP_OK);
    CHECK(t_back[10] == (float)stamps[10] && t_back.back() == (float)stamps.back());
    auto copy = PointCloudGPU::clone(both);
    CHECK(copy->size() == both.size() && copy->has_points() && copy->has_points_gpu() && copy->has_covs_gpu() && copy->has_times_gpu());
    CHECK(copy->points_gpu != both.points_gpu && copy->points != both.points);  // deep copy
    const auto p0 = download_points_gpu(both), p1 = download_points_gpu(*copy);
    const auto c0 = download_covs_gpu(both), c1 = download_covs_gpu(*copy);
    bool same = p0.size() == p1.size() && c0.size() == c1.size();
    for (size_t i = 0; same && i < p0.size(); i += 97) same = p0[i](0) == p1[i](0) && p0[i](2) == p1[i](2) && c0[i](2, 1) == c1[i](2, 1);
    CHECK(same);
    // a frame that lives on the device only (no host attributes): clone() copies device to device, download_points() fills the CPU side
    auto dev_only = std::make_shared<PointCloudGPU>();
    dev_only->add_points_gpu(sp);
    dev_only->add_covs_gpu(sc);
    CHECK(!dev_only->has_points() && dev_only->has_points_gpu());
    auto copy2 = PointCloudGPU::clone(*dev_only);
    CHECK(copy2->has_points_gpu() && copy2->has_covs_gpu() && copy2->points_gpu != dev_only->points_gpu);
    const auto p2 = download_points_gpu(*copy2);
    CHECK(p2.size() == sp.size() && p2[123](1) == sp[123](1));
    copy2->download_points();
    CHECK(copy2->has_points() && copy2->points[123](1) == (double)sp[123](1) && copy2->points[123](3) == 1.0);
    CHECK(copy2->offload_gpu() && !copy2->loaded_on_gpu() && copy2->reload_gpu() && copy2->has_covs_gpu());  // offload / reload of a cloned frame
  }

  auto voxels = std::make_shared<GaussianVoxelMapGPU>(0.5f);
  voxels->insert(*target);
  CHECK(voxels->voxelmap_info.num_voxels > 100 && voxels->buckets != nullptr && voxels->loaded_on_gpu());
  const auto means = download_voxel_means(*voxels);
  const auto buckets = download_buckets(*voxels);
  CHECK((int)means.size() == voxels->voxelmap_info.num_voxels && (int)buckets.size() == voxels->voxelmap_info.num_buckets);
  int used = 0;
  for (const auto& b : buckets) used += b.second >= 0;
  CHECK(used == voxels->voxelmap_info.num_voxels);  // test_voxelmap.cpp:352-380
  const double self_overlap = overlap_gpu(voxels, target, I_iso);
  CHECK(self_overlap > 0.99);  // test_voxelmap.cpp:226

  // save / load round trip
  voxels->save_compact("/tmp/gp_host_voxels.bin");
  auto loaded = GaussianVoxelMapGPU::load("/tmp/gp_host_voxels.bin");
  CHECK(loaded && loaded->voxelmap_info.num_voxels == voxels->voxelmap_info.num_voxels);
  CHECK(std::fabs(overlap_gpu(loaded, source, T_iso) - overlap_gpu(voxels, source, T_iso)) < 1e-3);

  // the overload set of overlap_gpu (types/gaussian_voxelmap.hpp:72-165) + merge_frames_gpu
  {
    const std::vector<GaussianVoxelMap::ConstPtr> two{voxels, loaded};
    const double single = overlap_gpu(voxels, source, T_iso);
    const double u = overlap_gpu(two, source, std::vector<Eigen::Isometry3d>{T_iso, T_iso});
    CHECK(std::fabs(u - single) < 1e-12);  // the same map twice: union == single
    const auto rates = overlap_gpu(two, std::vector<PointCloud::ConstPtr>{target, source}, std::vector<Eigen::Isometry3d>{I_iso, T_iso});
    CHECK(rates.size() == 2 && rates[0] == self_overlap && std::fabs(rates[1] - u) < 1e-12);
    // templated forwarding overloads of the reference header (vector of derived pointers)
    const std::vector<GaussianVoxelMapGPU::ConstPtr> two_gpu{voxels, loaded};
    CHECK(overlap_gpu(two_gpu, source, std::vector<Eigen::Isometry3d>{T_iso, T_iso}) == u);
    // pose resident in device memory as an Eigen::Isometry3f (:72-76)
    const Eigen::Isometry3f Tf = T_iso.cast<float>();
    void* d_pose = nullptr;
    CHECK(gp_malloc(&d_pose, sizeof(float) * 16) == GP_OK && gp_memcpy_h2d(d_pose, Tf.data(), sizeof(float) * 16, nullptr) == GP_OK && gp_stream_synchronize(nullptr) == GP_OK);
    const double from_dev = overlap_gpu(voxels, source, static_cast<const Eigen::Isometry3f*>(d_pose));
    CHECK(std::fabs(from_dev - single) < 1e-3);  // the float pose moves a handful of points across voxel faces
    gp_free(d_pose);
    // a point cloud as the target (:93-97)
    CHECK(overlap_gpu(std::static_pointer_cast<const PointCloud>(target), std::static_pointer_cast<const PointCloud>(source), T_iso) > 0.9);
    // double 4-vector / 4x4 inputs go through the same pack kernels
    std::vector<Eigen::Vector4d> p4((size_t)N);
    std::vector<Eigen::Matrix4d> c4((size_t)N, Eigen::Matrix4d::Zero());
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < 3; k++) p4[(size_t)i][k] = sp[(size_t)i][k];
      p4[(size_t)i][3] = 1.0;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) c4[(size_t)i](r, c) = sc[(size_t)i](r, c);
    }
    auto source4 = std::make_shared<PointCloudGPU>();
    source4->add_points_gpu(p4);
    source4->add_covs_gpu(c4);
    CHECK(overlap_gpu(voxels, source4, T_iso) == single);
    const auto back = download_points_gpu(*source4);
    CHECK(back.size() == (size_t)N && back[7][1] == sp[7][1]);
    auto merged = merge_frames_gpu({I_iso, T_iso}, std::vector<PointCloud::ConstPtr>{target, source4}, 0.25);
    CHECK(merged->size() > 1000 && merged->size() < (size_t)(2 * N) && merged->points_gpu && merged->covs_gpu);
    CHECK(overlap_gpu(voxels, merged, I_iso) > 0.9);  // both frames land on the target's surfaces
    // the merged frame is a complete frame like the reference's (add_points / add_covs / add_intensities, gaussian_voxelmap_gpu_funcs.cu:146-149):
    // CPU attributes present and equal to the device arrays
    CHECK(merged->has_points() && merged->has_covs() && merged->has_intensities());
    const auto mp = download_points_gpu(*merged);
    const auto mc = download_covs_gpu(*merged);
    CHECK(mp.size() == merged->size() && mc.size() == merged->size());
    bool same_host = true;
    for (size_t i = 0; i < merged->size(); i++) {
      for (int r = 0; r < 3; r++) same_host = same_host && merged->points[i](r) == (double)mp[i](r);
      same_host = same_host && merged->points[i](3) == 1.0 && merged->covs[i](3, 3) == 0.0;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) same_host = same_host && merged->covs[i](r, c) == (double)mc[i](r, c);
    }
    CHECK(same_host);
    // a null intensity source allocates zeros, like add_times_gpu
    PointCloudGPU blank;
    blank.add_points(sp);
    blank.add_intensities(static_cast<const float*>(nullptr), N);
    CHECK(blank.has_intensities() && blank.has_intensities_gpu() && blank.intensities[3] == 0.0);
    const auto zi = download_intensities_gpu(blank);
    CHECK(zi.size() == (size_t)N && zi[3] == 0.0f && zi.back() == 0.0f);
  }

  // OffloadableGPU round trip on a cloud (the reference's own touch() / access counter, types/offloadable.cpp)
  {
    const double before = overlap_gpu(voxels, source, T_iso);
    const auto t0 = OffloadableGPU::current_access_time();
    CHECK(source->loaded_on_gpu() && source->memory_usage_gpu() == (size_t)48 * N);
    CHECK(source->offload_gpu() && !source->offload_gpu() && !source->loaded_on_gpu() && source->points_gpu == nullptr);
    CHECK(source->touch() && source->loaded_on_gpu() && !source->reload_gpu());
    CHECK(OffloadableGPU::current_access_time() == t0 + 1 && source->last_accessed_time() == t0);
    CHECK(overlap_gpu(voxels, source, T_iso) == before);
    // make_sure_loaded_on_gpu (gaussian_voxelmap_gpu_funcs.cu:21-40): overlap_gpu itself brings offloaded operands back -- every overload
    CHECK(source->offload_gpu() && voxels->offload_gpu() && !source->loaded_on_gpu() && !voxels->loaded_on_gpu());
    CHECK(overlap_gpu(voxels, source, T_iso) == before && source->loaded_on_gpu() && voxels->loaded_on_gpu());
    std::vector<GaussianVoxelMap::ConstPtr> two{voxels, voxels};
    CHECK(source->offload_gpu() && voxels->offload_gpu());
    CHECK(std::fabs(overlap_gpu(two, source, std::vector<Eigen::Isometry3d>{T_iso, T_iso}) - before) < 1e-12 && source->loaded_on_gpu() && voxels->loaded_on_gpu());
    CHECK(source->offload_gpu() && voxels->offload_gpu());
    const auto r2 = overlap_gpu(two, std::vector<PointCloud::ConstPtr>{source, source}, std::vector<Eigen::Isometry3d>{T_iso, T_iso});
    CHECK(r2.size() == 2 && r2[0] == before && r2[1] == before && source->loaded_on_gpu());
    CHECK(source->offload_gpu());
    auto m2 = merge_frames_gpu({I_iso}, std::vector<PointCloud::ConstPtr>{source}, 0.25);
    CHECK(m2->size() > 500 && source->loaded_on_gpu());
  }

  // factor through the round-robin pool + the REFERENCE's linearisation hook, as the applications do
  LinearizationHook::register_hook([] { return create_nonlinear_factor_set_gpu(); });
  StreamTempBufferRoundRobin roundrobin(4);
  auto sb = roundrobin.get_stream_buffer();
  gtsam::NonlinearFactorGraph graph;
  auto factor = graph.emplace_shared<IntegratedVGICPFactorGPU>(world, 1, voxels, source, sb.first, sb.second);  // unary: fixed target
  auto sb2 = roundrobin.get_stream_buffer();
  auto factor_b = graph.emplace_shared<IntegratedVGICPFactorGPU>(0, 1, voxels, source, sb2.first, sb2.second);  // binary
  LinearizationHook hook(graph);
  CHECK(hook.size() == 2);
  CHECK((factor->get_fixed_target_pose().matrix() - Eigen::Matrix4f::Identity()).norm() == 0.0f && factor_b->get_target() == voxels);

  gtsam::Values values;
  values.insert(0, world);
  values.insert(1, T_true * gtsam::Pose3::Expmap(xi6(0.03, -0.02, 0.04, 0.1, 0.08, -0.05)));
  double lambda = 1e-5, err = 0.0;
  for (int iter = 0; iter < 30; iter++) {
    hook.linearize(values);
    auto hf = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor->linearize(values));
    auto hb = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor_b->linearize(values));
    CHECK(hf && hb && hb->binary);
    // the binary factor's source block equals the unary factor's (same delta): batch consistency
    for (int k = 0; k < 36; k++) CHECK(std::fabs(hb->G22.data()[k] - hf->G22.data()[k]) <= 1e-9 * (1.0 + std::fabs(hf->G22.data()[k])));
    err = hf->f;
    bool improved = false;
    for (int t = 0; t < 10 && !improved; t++) {
      gtsam::Matrix6 A = hf->G22;
      for (int d = 0; d < 6; d++) A(d, d) *= (1.0 + lambda);
      gtsam::Vector6 dx = gtsam::Vector6::Zero();
      CHECK(solve6(A, hf->g2, dx));
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
  const Eigen::Matrix4d d = (T_true.inverse() * values.at<gtsam::Pose3>(1)).matrix();
  const double trace = d(0, 0) + d(1, 1) + d(2, 2);
  const double ang = std::acos(std::min(1.0, std::max(-1.0, (trace - 1.0) / 2.0)));
  const double trans = std::sqrt(d(0, 3) * d(0, 3) + d(1, 3) * d(1, 3) + d(2, 3) * d(2, 3));
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
  std::shared_ptr<gtsam::HessianFactor> ref_lin;
  {
    hook.linearize(values);
    ref_lin = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor_b->linearize(values));
    (void)factor->linearize(values);
    factor->set_enable_offloading(true);
    factor_b->set_enable_offloading(true);
    CHECK(voxels->offload_gpu() && source->offload_gpu());
    hook.linearize(values);
    auto again = std::dynamic_pointer_cast<gtsam::HessianFactor>(factor_b->linearize(values));
    (void)factor->linearize(values);
    CHECK(voxels->loaded_on_gpu() && source->loaded_on_gpu());
    for (int k = 0; k < 36; k++)
      CHECK(again->G11.data()[k] == ref_lin->G11.data()[k] && again->G22.data()[k] == ref_lin->G22.data()[k] && again->G12.data()[k] == ref_lin->G12.data()[k]);
    CHECK(again->f == ref_lin->f);
  }
  // a third-party NonlinearFactorGPU goes through the generic staging-buffer path of the set and gives the same record
  {
    auto inner = std::make_shared<IntegratedVGICPFactorGPU>(0, 1, voxels, source);
    auto set = create_nonlinear_factor_set_gpu();
    CHECK(set->add(std::make_shared<WrappedFactor>(inner)) && set->size() == 1);
    auto lin = set->calc_linear_factors(values);
    auto h = std::dynamic_pointer_cast<gtsam::HessianFactor>(lin[0]);
    CHECK(h && h->f == ref_lin->f);
    for (int k = 0; k < 36; k++) CHECK(h->G22.data()[k] == ref_lin->G22.data()[k]);
  }
  // the sharded set: four factors in two shards (on one GPU: two shards on the same device with the host gather; on a multi-GPU
  // node the same call spreads shards over devices and all-reduces the stacked records with RCCL) == the single batch
  {
    std::vector<std::shared_ptr<IntegratedVGICPFactorGPU>> fs;
    NonlinearFactorSetGPU single, sharded;
    for (int k = 0; k < 4; k++) {
      fs.push_back(std::make_shared<IntegratedVGICPFactorGPU>(0, 1, k % 2 ? loaded : voxels, source));
      single.add(fs.back());
    }
    auto a = single.calc_linear_factors(values);
    CHECK(single.num_shards() == 1);
    std::vector<std::shared_ptr<gtsam::HessianFactor>> ref;
    for (auto& g : a) ref.push_back(std::dynamic_pointer_cast<gtsam::HessianFactor>(g));
    for (auto& f : fs) sharded.add(f);
    sharded.set_shard_assignment({0, 1, 0, 1}, 2);
    auto b = sharded.calc_linear_factors(values);
    CHECK(sharded.num_shards() == 2 && !sharded.uses_rccl());
    for (int k = 0; k < 4; k++) {
      auto hb2 = std::dynamic_pointer_cast<gtsam::HessianFactor>(b[(size_t)k]);
      CHECK(hb2 && hb2->f == ref[(size_t)k]->f);
      for (int e = 0; e < 36; e++) CHECK(hb2->G12.data()[e] == ref[(size_t)k]->G12.data()[e]);
    }
    sharded.error(values);
    for (auto& f : fs) CHECK(f->error(values) > 0.0);
    // the partitioner of the C-ABI: 10 equal factors over 4 shards
    int64_t w[10];
    for (auto& x : w) x = 32768;
    gp_shard_plan_t* plan = nullptr;
    CHECK(gp_shard_plan_create(w, 10, 4, &plan) == GP_OK && gp_shard_plan_num_shards(plan) == 4);
    int covered = 0, worst = 0;
    for (int s = 0; s < 4; s++) {
      int b0 = 0, e0 = 0;
      CHECK(gp_shard_plan_range(plan, s, &b0, &e0) == GP_OK && b0 == covered && e0 > b0);
      covered = e0;
      worst = std::max(worst, e0 - b0);
    }
    CHECK(covered == 10 && worst == 3);
    gp_shard_plan_destroy(plan);
  }
  // the ISAM2Ext cadence through the REFERENCE's LinearizationHook (isam2_ext.cpp; the optimizer itself needs GTSAM's ISAM2 and is not compiled here):
  //   :52        a default-constructed hook (one set per registered factory), empty
  //   :88-129    clear(); per candidate factor add(factor) -> true: a GPU factor, kept for the batched pass; false: the caller linearises it on the host;
  //              then calc_linear_factors(theta) returns the GPU factors' Hessians IN THE ORDER OF THE ADDS
  //   :434       clear_counts()
  //   :451-454   clear(); add(graph); linearize(theta); error(estimate)   (:236-238, :480-482, :494-497 are the same calls)
  //   :504-505   linearization_count() / evaluation_count() into the result
  {
    LinearizationHook ihook;
    CHECK(ihook.size() == 0);
    ihook.clear();
    gtsam::NonlinearFactor::shared_ptr host_only = std::make_shared<HostOnlyFactor>();
    std::vector<int> gpu_indices;
    std::vector<gtsam::NonlinearFactor::shared_ptr> candidates = {factor_b, host_only, factor};
    for (int idx = 0; idx < 3; idx++)
      if (ihook.add(candidates[(size_t)idx])) gpu_indices.push_back(idx);
    CHECK(gpu_indices.size() == 2 && gpu_indices[0] == 0 && gpu_indices[1] == 2 && ihook.size() == 2);  // the host-only factor was declined
    auto lin = ihook.calc_linear_factors(values);
    CHECK(lin.size() == 2);
    auto h0 = std::dynamic_pointer_cast<gtsam::HessianFactor>(lin[0]);
    auto h1 = std::dynamic_pointer_cast<gtsam::HessianFactor>(lin[1]);
    CHECK(h0 && h1 && h0->binary && !h1->binary);  // add order: the binary factor first, then the unary one
    CHECK(h0->f == ref_lin->f);
    for (int k = 0; k < 36; k++) CHECK(h0->G11.data()[k] == ref_lin->G11.data()[k] && h0->G22.data()[k] == ref_lin->G22.data()[k] && h0->G12.data()[k] == ref_lin->G12.data()[k]);
    CHECK(ihook.linearization_count() > 0);
    ihook.clear_counts();
    CHECK(ihook.linearization_count() == 0 && ihook.evaluation_count() == 0);
    ihook.clear();
    CHECK(ihook.size() == 0);
    gtsam::NonlinearFactorGraph all;
    all.push_back(factor);
    all.push_back(host_only);
    all.push_back(factor_b);
    ihook.add(all);
    CHECK(ihook.size() == 2);
    ihook.linearize(values);
    ihook.error(values);
    CHECK(ihook.linearization_count() == 2 && ihook.evaluation_count() == 2);
    const double e_cached = factor_b->error(values);  // (consumes the value the batched pass left: integrated_vgicp_factor_gpu.cpp:166-170)
    CHECK(std::fabs(e_cached - ref_lin->f) <= 1e-9 * std::fabs(ref_lin->f));
    (void)factor->error(values);
    (void)factor->linearize(values);
    (void)factor_b->linearize(values);
  }
  // the optimizer's trial with the values in device memory (levenberg_marquardt_graph_gpu.hpp over gp_lm_graph_*): LevenbergMarquardtOptimizerExt's calls, one by one --
  //   iterate() :352-392  linearize(values)                                   -> lm.set_values(values); lm.linearize()
  //   tryLambda() :188-350  buildDampedSystem + solve + retract + error(new)  -> lm.try_lambda(lambda, &dx, &b, &c, &new_error), then lm.accept()
  //   optimize() :394-430                                                     -> lm.optimize()
  // over a graph of the binary factor (0 -> 1), pose 0 held, and the unary factor (fixed world pose -> 1)
  {
    roundrobin.sync_all();
    auto f_bin = std::make_shared<IntegratedVGICPFactorGPU>(0, 1, voxels, source);
    auto f_un = std::make_shared<IntegratedVGICPFactorGPU>(world, 1, voxels, source);
    gtsam::Values start;  // (`values` has been optimised by the loops above: start again from the perturbed pose)
    start.insert(0, world);
    start.insert(1, T_true * gtsam::Pose3::Expmap(xi6(0.03, -0.02, 0.04, 0.1, 0.08, -0.05)));
    LevenbergMarquardtGraphGPU lm({f_bin, f_un}, {0});
    CHECK(lm.dim() == 6 && lm.ordered_keys().size() == 2);
    lm.set_values(start);
    lm.linearize();
    std::vector<double> dx, bvec;
    double c = 0.0, e_new = 0.0;
    CHECK(lm.try_lambda(1e-5, &dx, &bvec, &c, &e_new));
    auto h_bin = std::dynamic_pointer_cast<gtsam::HessianFactor>(f_bin->linearize(start));
    auto h_un = std::dynamic_pointer_cast<gtsam::HessianFactor>(f_un->linearize(start));
    CHECK(std::fabs(c - (h_bin->f + h_un->f)) <= 1e-9 * c);  // the cost at the linearisation point = the two factors' own
    CHECK(dx.size() == 6 && e_new < c);                      // the step decreases the cost on the frozen correspondences
    // the trial's error == the factors' own error() at the retracted values (pose 1 <- pose 1 * Expmap(dx))
    gtsam::Values trial;
    trial.insert(0, world);
    trial.insert(1, start.at<gtsam::Pose3>(1).retract(xi6(dx[0], dx[1], dx[2], dx[3], dx[4], dx[5])));
    const double e_host = f_bin->error(trial) + f_un->error(trial);
    CHECK(std::fabs(e_new - e_host) <= 1e-8 * e_host);
    lm.accept();
    const gtsam::Values after_one = lm.values();
    CHECK((after_one.at<gtsam::Pose3>(1).matrix() - trial.at<gtsam::Pose3>(1).matrix()).norm() < 1e-12 && (after_one.at<gtsam::Pose3>(0).matrix() - world.matrix()).norm() == 0.0);
    // the library's loop from the start values: ends at the generator's pose
    lm.set_values(start);
    const gp_lm_summary sum = lm.optimize();
    std::printf("lm graph: %d iterations (%d trials), final error %.6g (start %.6g, first trial %.6g), lambda %.3g\n", sum.iterations, sum.inner_iterations, sum.final_error, c, e_new, sum.final_lambda);
    // (final_error is measured on the LAST linearisation's correspondences, e_new on the first one's: they are not comparable; the start cost is an upper bound of both)
    CHECK(sum.iterations >= 2 && sum.inner_iterations >= sum.iterations && !sum.gave_up && sum.final_error < c);
    const Eigen::Matrix4d E = T_true.inverse().matrix() * lm.values().at<gtsam::Pose3>(1).matrix();
    const double ang = std::acos(std::min(1.0, std::max(-1.0, (E.block<3, 3>(0, 0).trace() - 1.0) / 2.0))), tr = E.block<3, 1>(0, 3).norm();
    CHECK(ang < 0.015 && tr < 0.15);  // test_matching_cost_factors.cpp:227
  }
  roundrobin.sync_all();
  std::printf("HOST_TEST_OK\n");
  return 0;
}

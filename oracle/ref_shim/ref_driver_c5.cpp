// ref_driver_c5.cpp -- C entry points around the REFERENCE's own CPU code of config 5: KdTree (ann/kdtree.cpp over
// ann/small_kdtree.hpp + knn_result.hpp), estimate_covariances (features/covariance_estimation.cpp) and
// IntegratedGICPFactor_<PointCloud, PointCloud> (factors/impl/integrated_gicp_factor_impl.hpp), compiled from
// /root/reference where they lie against the stand-in headers of ./include.  TEST INFRASTRUCTURE ONLY (oracle/_ref/libref.so).
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include <gtsam_points/ann/kdtree.hpp>
#include <gtsam_points/features/covariance_estimation.hpp>
#include <gtsam_points/types/point_cloud.hpp>
#include <gtsam_points/factors/integrated_gicp_factor.hpp>
#include <gtsam_points/factors/impl/integrated_gicp_factor_impl.hpp>
#include <gtsam/linear/HessianFactor.h>

// the instantiation src/gtsam_points/factors/integrated_gicp_factor.cpp:10 makes (that file also instantiates the iVox variants,
// which would drag in the incremental voxel ANN sources)
template class gtsam_points::IntegratedGICPFactor_<gtsam_points::PointCloud, gtsam_points::PointCloud>;

namespace {

struct Cloud4 : public gtsam_points::PointCloud {
  std::vector<Eigen::Vector4d> pts;
  std::vector<Eigen::Matrix4d> cvs;
  Cloud4(const float* p, const float* c, int n) : pts(n), cvs(c ? n : 0) {
    for (int i = 0; i < n; i++) {
      pts[i] = Eigen::Vector4d(p[3 * i], p[3 * i + 1], p[3 * i + 2], 1.0);
      if (c) {
        cvs[i].setZero();
        for (int cc = 0; cc < 3; cc++)
          for (int r = 0; r < 3; r++) cvs[i](r, cc) = c[9 * i + cc * 3 + r];
      }
    }
    num_points = n;
    points = pts.data();
    covs = c ? cvs.data() : nullptr;
  }
};

struct RefTree {
  std::vector<Eigen::Vector4d> pts;
  std::unique_ptr<gtsam_points::KdTree> tree;
};

struct RefGICP {
  std::shared_ptr<Cloud4> target, source;
  std::shared_ptr<gtsam_points::IntegratedGICPFactor> factor;
};

gtsam::Values values_of(const double* delta) {
  Eigen::Matrix4d m;
  std::memcpy(m.data(), delta, sizeof(double) * 16);
  gtsam::Values v;
  v.insert(0, gtsam::Pose3());
  v.insert(1, gtsam::Pose3(m));
  return v;
}

}  // namespace

struct ref_linearized6 {
  int num_inliers;
  int pad_;
  double error;
  double H_target[36], H_source[36], H_target_source[36], b_target[6], b_source[6];
};

extern "C" {

void* ref_kdtree_create(const float* points, int n) {
  auto* t = new RefTree;
  t->pts.resize(n);
  for (int i = 0; i < n; i++) t->pts[i] = Eigen::Vector4d(points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.0);
  t->tree.reset(new gtsam_points::KdTree(t->pts.data(), n, 1));
  return t;
}
void ref_kdtree_destroy(void* h) { delete static_cast<RefTree*>(h); }
int ref_kdtree_knn(void* h, const double* query3, int k, long long* indices, double* sq_dists, double max_sq_dist) {
  const double q[4] = {query3[0], query3[1], query3[2], 1.0};
  std::vector<size_t> idx(k);
  const size_t found = static_cast<RefTree*>(h)->tree->knn_search(q, (size_t)k, idx.data(), sq_dists, max_sq_dist);
  for (size_t i = 0; i < found; i++) indices[i] = (long long)idx[i];
  return (int)found;
}

// estimate_covariances(points, n, k, num_threads), covariance_estimation.cpp:79-85 (EIG regularisation, eigenvalues 1e-3,1,1);
// out: double[n][9] column-major top-left 3x3
void ref_estimate_covariances(const float* points, int n, int k, int num_threads, double* out) {
  std::vector<Eigen::Vector4d> pts(n);
  for (int i = 0; i < n; i++) pts[i] = Eigen::Vector4d(points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.0);
  const auto covs = gtsam_points::estimate_covariances(pts.data(), n, k, num_threads);
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) out[9 * (size_t)i + 3 * c + r] = covs[i](r, c);
}

void* ref_gicp_create(const float* tpoints, const float* tcovs, int nt, const float* points, const float* covs, int n, int num_threads, double max_corr_dist_sq) {
  auto* f = new RefGICP;
  f->target = std::make_shared<Cloud4>(tpoints, tcovs, nt);
  f->source = std::make_shared<Cloud4>(points, covs, n);
  f->factor = std::make_shared<gtsam_points::IntegratedGICPFactor>(0, 1, f->target, f->source);
  f->factor->set_num_threads(num_threads);
  f->factor->set_max_correspondence_distance(std::sqrt(max_corr_dist_sq));
  return f;
}
void ref_gicp_destroy(void* h) { delete static_cast<RefGICP*>(h); }
void ref_gicp_linearize(void* h, const double* delta, ref_linearized6* out) {
  auto* f = static_cast<RefGICP*>(h);
  auto lin = std::dynamic_pointer_cast<gtsam::HessianFactor>(f->factor->linearize(values_of(delta)));
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out->H_target, lin->G11.data(), sizeof(double) * 36);
  std::memcpy(out->H_target_source, lin->G12.data(), sizeof(double) * 36);
  std::memcpy(out->H_source, lin->G22.data(), sizeof(double) * 36);
  for (int i = 0; i < 6; i++) {
    out->b_target[i] = -lin->g1[i];
    out->b_source[i] = -lin->g2[i];
  }
  out->error = lin->f;
  out->num_inliers = (int)std::lround(f->factor->inlier_fraction() * (double)f->source->size());  // integrated_gicp_factor.hpp:112-116
}

}  // extern "C"

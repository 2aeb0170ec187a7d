// ref_driver.cpp -- C entry points around the REFERENCE's own CPU VGICP classes (compiled from /root/reference where they
// lie, against the stand-in Eigen/GTSAM headers of ./include).  TEST INFRASTRUCTURE ONLY: builds oracle/_ref/libref.so,
// which pins oracle/vgicp_oracle.c (tests/test_ref_pin_cpu.py) and can serve as cpu_baseline kind "reference".
#include <cstring>
#include <memory>
#include <vector>

#include <gtsam_points/factors/integrated_vgicp_factor.hpp>
#include <gtsam_points/types/gaussian_voxelmap_cpu.hpp>
#include <gtsam_points/types/point_cloud.hpp>
#include <gtsam/linear/HessianFactor.h>

#include <gtsam_points/types/point_cloud_cpu.hpp>
// IncrementalVoxelMap<>::voxel_data() (a virtual that this path never calls) instantiates PointCloudCPU.  Its default
// constructor / destructor are empty bodies in src/gtsam_points/types/point_cloud_cpu.cpp:29-31, a file that also drags in the sampling /
// kd-tree utilities; the same empty body is provided here instead of compiling that file.
namespace gtsam_points {
PointCloudCPU::PointCloudCPU() {}
PointCloudCPU::~PointCloudCPU() {}  // :31 (key function: emits the vtable)
}  // namespace gtsam_points

namespace {

struct OwnedCloud : public gtsam_points::PointCloud {
  std::vector<Eigen::Vector4d> pts;
  std::vector<Eigen::Matrix4d> cvs;
  OwnedCloud(const float* p, const float* c, int n) : pts(n), cvs(n) {
    // what PointCloudCPU::add_points / add_covs do for float input: (x, y, z, 1) and a 4x4 with zero 4th row/column
    for (int i = 0; i < n; i++) {
      pts[i] = Eigen::Vector4d(p[3 * i], p[3 * i + 1], p[3 * i + 2], 1.0);
      cvs[i].setZero();
      for (int cc = 0; cc < 3; cc++)
        for (int r = 0; r < 3; r++) cvs[i](r, cc) = c[9 * i + cc * 3 + r];
    }
    num_points = n;
    points = pts.data();
    covs = cvs.data();
  }
};

struct RefFactor {
  std::shared_ptr<OwnedCloud> source;
  std::shared_ptr<gtsam_points::IntegratedVGICPFactor> factor;
};

gtsam::Values make_values(const double* delta) {
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

void* ref_voxelmap_create(double resolution) { return new std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>(new gtsam_points::GaussianVoxelMapCPU(resolution)); }
void ref_voxelmap_destroy(void* h) { delete static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h); }
void ref_voxelmap_insert(void* h, const float* points, const float* covs, int n) {
  OwnedCloud cloud(points, covs, n);
  (*static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h))->insert(cloud);
}
int ref_voxelmap_num_voxels(void* h) { return (int)(*static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h))->num_voxels(); }

// ---- on-disk format interop (src/test/test_voxelmap.cpp:301-432): the reference's own save_compact / load ----
void ref_voxelmap_save_compact(void* h, const char* path) { (*static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h))->save_compact(path); }
void* ref_voxelmap_load(const char* path) {
  auto m = gtsam_points::GaussianVoxelMapCPU::load(path);
  return m ? new std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>(m) : nullptr;
}
double ref_voxelmap_resolution(void* h) { return (*static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h))->voxel_resolution(); }
// voxel statistics in the map's own order: coords int[V][3] (voxel_coord of the mean), num_points int[V], means double[V][3],
// covs double[V][9] column-major, intensities double[V]
void ref_voxelmap_export(void* h, int* coords, int* num_points, double* means, double* covs, double* intensities) {
  auto& m = *static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h);
  const int V = (int)m->num_voxels();
  for (int v = 0; v < V; v++) {
    const auto& vox = m->lookup_voxel(v);
    const Eigen::Vector3i c = m->voxel_coord(vox.mean);
    for (int k = 0; k < 3; k++) {
      coords[3 * v + k] = c[k];
      means[3 * v + k] = vox.mean[k];
    }
    num_points[v] = (int)vox.num_points;
    for (int cc = 0; cc < 3; cc++)
      for (int r = 0; r < 3; r++) covs[9 * v + cc * 3 + r] = vox.cov(r, cc);
    intensities[v] = vox.intensity;
  }
}
// overlap(target, source, T) (src/gtsam_points/types/gaussian_voxelmap_cpu_funcs.cpp:126-143): the loop of that function over the
// reference's own voxel_coord / lookup_voxel_index (the file itself also holds merge_frames and needs PointCloudCPU's samplers)
double ref_voxelmap_overlap(void* h, const float* points, int n, const double* delta) {
  auto& m = *static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(h);
  Eigen::Matrix4d T;
  std::memcpy(T.data(), delta, sizeof(double) * 16);
  int num_overlap = 0;
  for (int i = 0; i < n; i++) {
    const Eigen::Vector4d pt = T * Eigen::Vector4d(points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.0);
    if (m->lookup_voxel_index(m->voxel_coord(pt)) >= 0) num_overlap++;
  }
  return n ? static_cast<double>(num_overlap) / n : 0.0;
}

void* ref_vgicp_create(void* map, const float* points, const float* covs, int n, int num_threads) {
  auto* f = new RefFactor;
  f->source = std::make_shared<OwnedCloud>(points, covs, n);
  auto voxels = *static_cast<std::shared_ptr<gtsam_points::GaussianVoxelMapCPU>*>(map);
  f->factor = std::make_shared<gtsam_points::IntegratedVGICPFactor>(0, 1, voxels, f->source);
  f->factor->set_num_threads(num_threads);
  return f;
}
void ref_vgicp_destroy(void* h) { delete static_cast<RefFactor*>(h); }

void ref_vgicp_linearize(void* h, const double* delta, ref_linearized6* out) {
  auto* f = static_cast<RefFactor*>(h);
  auto lin = std::dynamic_pointer_cast<gtsam::HessianFactor>(f->factor->linearize(make_values(delta)));
  std::memset(out, 0, sizeof(*out));
  // HessianFactor(k0, k1, H_t, H_ts, -b_t, H_s, -b_s, err), integrated_matching_cost_factor.cpp:49
  std::memcpy(out->H_target, lin->G11.data(), sizeof(double) * 36);
  std::memcpy(out->H_target_source, lin->G12.data(), sizeof(double) * 36);
  std::memcpy(out->H_source, lin->G22.data(), sizeof(double) * 36);
  for (int i = 0; i < 6; i++) {
    out->b_target[i] = -lin->g1[i];
    out->b_source[i] = -lin->g2[i];
  }
  out->error = lin->f;
  out->num_inliers = f->factor->num_inliers();
}

double ref_vgicp_error(void* h, const double* delta) { return static_cast<RefFactor*>(h)->factor->error(make_values(delta)); }

}  // extern "C"

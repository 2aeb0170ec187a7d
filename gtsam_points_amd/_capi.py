"""ctypes binding of libgtsam_points_hip.so (include/gtsam_points_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing
it raises.  torch is imported first so that the process shares ONE HIP runtime (the library's
NEEDED libamdhip64.so.7 resolves to the copy torch already loaded).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgtsam_points_hip.so")
_LIB = None


class GPError(RuntimeError):
    pass


class VoxelMapInfo(C.Structure):
    """VoxelMapInfo, types/gaussian_voxelmap_gpu.hpp:20-25"""

    _fields_ = [("num_voxels", C.c_int), ("num_buckets", C.c_int), ("max_bucket_scan_count", C.c_int), ("voxel_resolution", C.c_float)]


class VoxelMapViews(C.Structure):
    _fields_ = [
        ("buckets", C.c_void_p),
        ("num_points", C.c_void_p),
        ("voxel_means", C.c_void_p),
        ("voxel_covs", C.c_void_p),
        ("voxel_intensities", C.c_void_p),
    ]


class Linearized6(C.Structure):
    """gp_linearized6: LinearizedSystem6 (cuda/kernels/linearized_system.cuh:10-71) in double."""

    _fields_ = [
        ("num_inliers", C.c_double),
        ("error", C.c_double),
        ("H_target", C.c_double * 36),
        ("H_source", C.c_double * 36),
        ("H_target_source", C.c_double * 36),
        ("b_target", C.c_double * 6),
        ("b_source", C.c_double * 6),
    ]


LINEARIZED6_DOUBLES = 122
assert C.sizeof(Linearized6) == LINEARIZED6_DOUBLES * 8

_SIGNATURES = {
    # runtime
    "gp_last_error": (C.c_char_p, []),
    "gp_version": (C.c_char_p, []),
    "gp_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gp_set_device": (C.c_int, [C.c_int]),
    "gp_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "gp_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "gp_device_synchronize": (C.c_int, []),
    "gp_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gp_stream_destroy": (C.c_int, [C.c_void_p]),
    "gp_stream_synchronize": (C.c_int, [C.c_void_p]),
    "gp_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gp_free": (C.c_int, [C.c_void_p]),
    "gp_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gp_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gp_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gp_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "gp_host_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gp_host_free": (C.c_int, [C.c_void_p]),
    # temp buffer / stream pool
    "gp_temp_buffer_create": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "gp_temp_buffer_get": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gp_temp_buffer_clear": (C.c_int, [C.c_void_p]),
    "gp_temp_buffer_clear_all": (C.c_int, [C.c_void_p]),
    "gp_temp_buffer_destroy": (C.c_int, [C.c_void_p]),
    "gp_stream_pool_create": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gp_stream_pool_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "gp_stream_pool_sync_all": (C.c_int, [C.c_void_p]),
    "gp_stream_pool_clear": (C.c_int, [C.c_void_p]),
    "gp_stream_pool_clear_all": (C.c_int, [C.c_void_p]),
    "gp_stream_pool_destroy": (C.c_int, [C.c_void_p]),
    # voxel map
    "gp_voxelmap_create": (C.c_int, [C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_voxelmap_destroy": (C.c_int, [C.c_void_p]),
    "gp_voxelmap_insert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "gp_voxelmap_info_get": (C.c_int, [C.c_void_p, C.POINTER(VoxelMapInfo)]),
    "gp_voxelmap_resolution": (C.c_double, [C.c_void_p]),
    "gp_voxelmap_views_get": (C.c_int, [C.c_void_p, C.POINTER(VoxelMapViews)]),
    "gp_voxelmap_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_voxelmap_download_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_voxelmap_assign": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_voxelmap_save_compact": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gp_voxelmap_load": (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_voxelmap_memory_usage_gpu": (C.c_size_t, [C.c_void_p]),
    "gp_voxelmap_loaded_on_gpu": (C.c_int, [C.c_void_p]),
    "gp_voxelmap_has_block_grid": (C.c_int, [C.c_void_p]),
    "gp_voxelmap_offload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_voxelmap_reload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_voxelmap_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
    "gp_voxelmap_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]),
    "gp_voxelmap_overlap_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "gp_voxelmap_overlap_batch": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    # callers either side of the path: merge_frames_gpu, PointCloudGPU upload
    "gp_transform_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_merge_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_double,
                                  C.c_double, C.c_void_p, C.POINTER(C.c_void_p)]),
    # dense normal equations on the device
    "gp_dense_system_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_dense_system_destroy": (C.c_int, [C.c_void_p]),
    "gp_dense_system_size": (C.c_int, [C.c_void_p]),
    "gp_dense_system_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "gp_dense_system_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_dense_system_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_dense_system_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_sparse_system_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_sparse_system_destroy": (C.c_int, [C.c_void_p]),
    "gp_sparse_system_size": (C.c_int, [C.c_void_p]),
    "gp_sparse_system_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gp_sparse_system_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "gp_sparse_system_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_sparse_system_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_sparse_system_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_sparse_system_set_one_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "gp_debug_sparse_step_trace": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_sparse_symbolic": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gp_sparse_symbolic_schedule": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gp_cloud_upload_vec3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gp_cloud_upload_mat3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    # factor
    "gp_vgicp_factor_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_vgicp_factor_destroy": (C.c_int, [C.c_void_p]),
    "gp_vgicp_factor_device": (C.c_int, [C.c_void_p]),
    # sharded batches (single process, several devices)
    "gp_shard_plan_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gp_shard_plan_num_shards": (C.c_int, [C.c_void_p]),
    "gp_shard_plan_range": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gp_shard_plan_destroy": (C.c_int, [C.c_void_p]),
    "gp_voxelmap_clone_to_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_vgicp_multi_batch_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gp_vgicp_multi_batch_destroy": (C.c_int, [C.c_void_p]),
    "gp_vgicp_multi_batch_size": (C.c_int, [C.c_void_p]),
    "gp_vgicp_multi_batch_num_shards": (C.c_int, [C.c_void_p]),
    "gp_vgicp_multi_batch_uses_rccl": (C.c_int, [C.c_void_p]),
    "gp_vgicp_multi_batch_shard_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "gp_vgicp_multi_batch_linearize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_multi_batch_compute_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_multi_batch_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gp_vgicp_factor_set_surface_validation": (C.c_int, [C.c_void_p, C.c_int]),
    "gp_vgicp_factor_set_source": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_factor_set_inlier_update_thresh": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "gp_vgicp_factor_num_points": (C.c_int, [C.c_void_p]),
    "gp_vgicp_factor_stream": (C.c_void_p, [C.c_void_p]),
    "gp_vgicp_linearization_input_size": (C.c_size_t, []),
    "gp_vgicp_linearization_output_size": (C.c_size_t, []),
    "gp_vgicp_evaluation_input_size": (C.c_size_t, []),
    "gp_vgicp_evaluation_output_size": (C.c_size_t, []),
    "gp_vgicp_factor_issue_linearize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_factor_issue_compute_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_factor_sync": (C.c_int, [C.c_void_p]),
    "gp_vgicp_factor_linearize": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(Linearized6)]),
    "gp_vgicp_factor_compute_error": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    # batch
    "gp_vgicp_batch_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_vgicp_batch_destroy": (C.c_int, [C.c_void_p]),
    "gp_vgicp_batch_size": (C.c_int, [C.c_void_p]),
    "gp_vgicp_batch_total_points": (C.c_int64, [C.c_void_p]),
    "gp_vgicp_batch_algorithmic_bytes": (C.c_int64, [C.c_void_p]),
    "gp_vgicp_batch_actual_bytes": (C.c_int64, [C.c_void_p]),
    "gp_source_mirror_invalidate": (C.c_int, [C.c_void_p]),
    "gp_source_mirror_bytes": (C.c_int64, []),
    "gp_vgicp_batch_issue_linearize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_issue_compute_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_sync": (C.c_int, [C.c_void_p]),
    "gp_vgicp_batch_linearize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_linearize_view": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_vgicp_batch_compute_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_point_grid_create": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_point_grid_destroy": (C.c_int, [C.c_void_p]),
    "gp_knn_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_estimate_covariances": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "gp_gicp_factor_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_gicp_factor_destroy": (C.c_int, [C.c_void_p]),
    "gp_gicp_factor_linearize": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(Linearized6)]),
    "gp_gicp_factor_compute_error": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gp_point_grid_create_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_estimate_covariances_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p]),
    "gp_gicp_factor_create_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    # per-handle tuning (no process-global switches)
    "gp_vgicp_batch_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gp_vgicp_batch_get_tuning": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gp_vgicp_batch_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gp_vgicp_batch_device_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gp_vgicp_factor_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gp_voxelmap_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gp_vgicp_batch_set_trace_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_debug_expand_rigid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_debug_stream_plan": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "gp_debug_multi_gather_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p]),
    "gp_peer_exchange_handle_bytes": (C.c_int, []),
    "gp_peer_exchange_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "gp_peer_exchange_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_peer_exchange_rows": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gp_peer_exchange_begin": (C.c_int, [C.c_void_p]),
    "gp_peer_exchange_set_timeout_ms": (C.c_int, [C.c_void_p, C.c_double]),
    "gp_peer_exchange_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_peer_exchange_check": (C.c_int, [C.c_void_p]),
    "gp_peer_exchange_destroy": (C.c_int, [C.c_void_p]),
    "gp_debug_side_stream_probe": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "gp_debug_inject_sort_fault": (C.c_int, [C.c_int]),
    "gp_debug_sort_fallbacks": (C.c_int, []),
    "gp_trim_device_cache": (C.c_int, []),
    "gp_vgicp_batch_issue_linearize_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "gp_vgicp_batch_issue_compute_error_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_compute_error_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_issue_compute_error_dev_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_compute_error_dev_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_dense_system_set_one_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "gp_dense_system_collect_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_sparse_system_collect_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_lm_graph_set_speculation": (C.c_int, [C.c_void_p, C.c_int]),
    "gp_lm_graph_set_one_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "gp_vgicp_batch_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gp_dense_system_issue_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "gp_dense_system_finish_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_dense_system_device_solution": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "gp_sparse_system_issue_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "gp_sparse_system_finish_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_sparse_system_device_solution": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    # the optimizer's trial with the values in device memory (gp_lm.hip)
    "gp_lm_params_default": (None, [C.c_void_p]),
    "gp_lm_graph_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "gp_lm_graph_destroy": (C.c_int, [C.c_void_p]),
    "gp_lm_graph_num_variables": (C.c_int, [C.c_void_p]),
    "gp_lm_graph_set_values": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_lm_graph_get_values": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gp_lm_graph_linearize": (C.c_int, [C.c_void_p]),
    "gp_lm_graph_try_lambda": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_lm_graph_accept": (C.c_int, [C.c_void_p]),
    "gp_lm_graph_optimize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_lm_graph_records": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "gp_debug_sparse_work_lists": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_vgicp_batch_time_linearize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}

class LmParams(C.Structure):
    """gp_lm_params (include/gtsam_points_hip.h): GTSAM's LevenbergMarquardtParams fields the loop reads"""

    _fields_ = [("lambda_initial", C.c_double), ("lambda_factor", C.c_double), ("lambda_upper_bound", C.c_double), ("lambda_lower_bound", C.c_double),
                ("relative_error_tol", C.c_double), ("absolute_error_tol", C.c_double), ("min_model_fidelity", C.c_double), ("min_diagonal", C.c_double),
                ("max_diagonal", C.c_double), ("max_iterations", C.c_int), ("diagonal_damping", C.c_int)]


class LmSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("inner_iterations", C.c_int), ("gave_up", C.c_int), ("reserved_", C.c_int), ("final_error", C.c_double), ("final_lambda", C.c_double)]


# tuning keys / kernel families of include/gtsam_points_hip.h
GP_KERNEL_REFERENCE, GP_KERNEL_HASHED, GP_KERNEL_GRID_F64, GP_KERNEL_LOOKAHEAD, GP_KERNEL_STREAM = 0, 2, 3, 8, 12
GP_TUNE_KERNEL, GP_TUNE_SOURCE_POLICY, GP_TUNE_XCD_CHUNK, GP_TUNE_STAGGER, GP_TUNE_TILE_INTERLEAVE, GP_TUNE_BALANCE, GP_TUNE_EFFECTIVE_KERNEL = 0, 1, 2, 3, 4, 5, 6
GP_TUNE_TIMING = 7
GP_TUNE_XCD_WEIGHT_0 = 8
GP_TUNE_FUSED_FINALIZE = 17
GP_TUNE_TILE_CHUNKS = 18
GP_TUNE_MAX_WORKGROUPS = 19
GP_TUNE_TEST_ARRIVAL_SKEW = 20
GP_TUNE_SOURCE_MIRROR, GP_TUNE_EFFECTIVE_MIRROR = 21, 22
GP_TUNE_EXPERIMENT = 23
GP_TUNE_BUCKET_LOAD = 24
GP_TUNE_MAP_BUILD, GP_TUNE_KNN_STRUCTURE = 16, 32
KERNEL_FAMILIES = [GP_KERNEL_REFERENCE, GP_KERNEL_HASHED, GP_KERNEL_GRID_F64, GP_KERNEL_LOOKAHEAD, GP_KERNEL_STREAM]

EXPORTED_SYMBOLS = sorted(list(_SIGNATURES.keys()) + ["gp_linearized6_to_f32"])


def load():
    """Load libgtsam_points_hip.so; raise loudly when it is absent (no CPU fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GPError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gtsam_points_amd/csrc`. gtsam_points_amd has no CPU fallback."
            )
        import torch  # noqa: F401  (one HIP runtime per process: torch's libamdhip64 first)

        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = lib
    return _LIB


_TUNE_SIGNATURES = {
    "gp_debug_stream_bench": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "gp_debug_calibration_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gp_debug_spin": (C.c_int, [C.c_double, C.c_void_p]),
    "gp_debug_sort_pairs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gp_debug_sort_pairs_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "gp_debug_occupy": (C.c_int, [C.c_double, C.c_int, C.c_void_p]),
}
_TUNE = None


def load_tune():
    """libgtsam_points_hip_tune.so: measurement kernels only (include/gtsam_points_hip_tune.h); never needed by the product path"""
    global _TUNE
    if _TUNE is None:
        load()
        path = os.path.join(_HERE, "libgtsam_points_hip_tune.so")
        if not os.path.exists(path):
            raise GPError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _TUNE_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _TUNE = lib
    return _TUNE


def check(rc, what=""):
    if rc != 0:
        msg = load().gp_last_error()
        raise GPError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

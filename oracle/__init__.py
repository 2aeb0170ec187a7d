"""CPU oracle for the VGICP path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (gtsam_points_amd/) never does.
"""
from .capi import (  # noqa: F401
    Linearized6,
    OracleGICPFactor,
    OracleKdTree,
    OracleVGICPFactor,
    OracleVoxelMap,
    build,
    calc_delta,
    estimate_covariances,
    expmap,
    max_threads,
    merge_frames,
)

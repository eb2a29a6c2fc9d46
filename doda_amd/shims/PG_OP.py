"""Drop-in `PG_OP` extension module for a DODA checkout."""
from doda_amd.pg_op import *  # noqa: F401,F403
from doda_amd.pg_op import (ballquery_batch_p, bfs_cluster, get_iou, knn_batch,  # noqa: F401
                            point_recover_bp, point_recover_fp, roipool_bp, roipool_fp, sec_max,
                            sec_mean, sec_mean_bp, sec_min, voxelize_bp, voxelize_fp, voxelize_idx)

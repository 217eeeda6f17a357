"""Farthest-point down-sampling (open3d PointCloud.farthest_point_down_sample, reference
cluster_icp.py:43).  SURVEY.md 8(f) row N1 -- scheduled after the hot path; not built this round."""


def farthest_point_sample(points, num_samples):
    raise NotImplementedError("Segments(sample_size=...) needs the FPS kernel (SURVEY.md 8f N1): not built yet")

"""Same import path, names, argument order and return convention as the reference's pybind11 module `shot`
(src_shot/shot.cpp:164-168): compute(pc, normal_r, shot_r) -> [float32[N*352], float32[N*3]] (shot.cpp:45-100),
estimate_normal(pc, normal_r) -> float32[N*3] (shot.cpp:12-42), compute_color(pc, rgb, normal_r, shot_r) -> float32[N*1344]
(shot.cpp:102-161).  Every call runs on the GPU through libcppf_hip.so (cppf_shot352 / cppf_estimate_normals / cppf_shot1344);
without the library or a HIP device the import / the call raises -- there is no CPU path."""
from cppf2_amd.shot import compute, compute_color, estimate_normal  # noqa: F401

__all__ = ["compute", "compute_color", "estimate_normal"]

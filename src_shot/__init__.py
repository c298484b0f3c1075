"""Package path of the reference's native module: `from src_shot.build import shot` (eval.py:15, dataset.py:12, train_dino.py:15).
The reference builds src_shot/shot.cpp (pybind11 + PCL) into src_shot/build/; here the same import resolves to the HIP library."""

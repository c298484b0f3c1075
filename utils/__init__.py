"""`utils.util` of the reference, for the names its entry points import (eval.py:5, train_shot.py:16, dataset.py:4,9,20)."""

"""`from src_shot.build import shot` -> src_shot/build/shot.py (the module the reference's CMake build drops here, shot.cpp:164-168)."""

"""`simple_knn` as a package around the COMPILED `_C` extension module (integration/torch_ext_pybind.cpp) [REF scene/gaussian_model.py:20]."""

"""gfx950 implementation of the reference's `simple_knn` CUDA extension (submodules/simple-knn): import as
`from simple_knn._C import distCUDA2` exactly like the reference does (scene/gaussian_model.py:20)."""

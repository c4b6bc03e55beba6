"""Drop-in for the `simple_knn` CUDA extension (scene/gaussian_model.py:20)."""

"""Drop-in import name of the reference's simple-knn extension
(``scene/gaussian_model.py:20``: ``from simple_knn._C import distCUDA2``).
Implemented in :mod:`gftorf_amd.knn`."""

"""``simple_knn._C`` of the reference (submodules/simple-knn/ext.cpp:15-17) on MI355X."""
from gftorf_amd.knn import distCUDA2  # noqa: F401

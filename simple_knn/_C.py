"""Drop-in for the reference's compiled `simple_knn._C` extension (`from simple_knn._C import distCUDA2`,
scene/gaussian_model.py:13)."""
from relightable3dgaussian_amd.knn_ops import distCUDA2  # noqa: F401

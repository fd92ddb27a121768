"""Package name the reference imports (`from simple_knn._C import distCUDA2`)."""

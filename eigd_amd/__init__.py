"""eigd_amd: MI355X-native implementation of smdogroup/eigd's adjoint eigenvector-derivative path."""
__version__ = "0.1.0"

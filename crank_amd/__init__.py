"""crank_amd: MI355X-native VQ-VAE voice-conversion training step (HIP/CDNA4 hot path
behind the Python API and YAML config surface of k2kobayashi/crank's trainers).

Importing the package does not load the HIP library; building a model or calling an op
does, and fails loudly if crank_amd/libcrank_hip.so is missing (no fallback path)."""
__version__ = "0.1.0"

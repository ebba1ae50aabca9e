"""visiondepth3d_amd -- MI355X-native depth-image-based stereo renderer (VisionDepth3D's per-frame hot path)."""
__version__ = "0.1.0"

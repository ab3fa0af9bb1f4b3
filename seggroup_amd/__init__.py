"""seggroup_amd: MI355X-native pseudo-label generation hot path of SegGroup."""
__version__ = "0.1.0"

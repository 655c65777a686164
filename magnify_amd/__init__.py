"""magnify_amd -- MI355X-native marker-detection hot path behind magnify's component API."""
__version__ = "0.1.0"

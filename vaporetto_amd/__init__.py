"""vaporetto_amd: MI355X-native boundary scoring for Vaporetto models."""
__version__ = "0.1.0"

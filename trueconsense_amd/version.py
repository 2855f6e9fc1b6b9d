__version__ = "0.1.0"
# behaviour tracks RIVM-bioinformatics/TrueConsense
REFERENCE_VERSION = "0.5.2"

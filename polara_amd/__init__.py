"""polara_amd: MI355X-native PureSVD / CoFFee hot path behind Polara's RecommenderModel surface."""
__version__ = '0.1.0'

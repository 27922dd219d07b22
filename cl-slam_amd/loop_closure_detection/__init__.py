"""Drop-in for the hot-path part of the reference's ``loop_closure_detection`` package: the feature
encoder forward (encoder.py).  The faiss search / bookkeeping of loop_closure_detection.py stays on
the host in the reference's own code."""
from loop_closure_detection.encoder import FeatureEncoder

__all__ = ['FeatureEncoder']

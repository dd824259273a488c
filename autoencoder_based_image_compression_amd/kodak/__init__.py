"""Host-side mirror of the reference's ``kodak_tensorflow/`` modules that sit on the hot path.

Same module names, function names, argument meaning and exceptions as the reference, so that parity tests read like
the reference's own tests and a driver shaped like ``reconstructing_eae_kodak.py`` runs on it.
"""

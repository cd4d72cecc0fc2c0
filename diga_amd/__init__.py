"""diga_amd -- MI355X-native implementation of the DiGA data-parallel training hot path.

Layout mirrors the reference tree `domain_adaptation/GTA5/` so that either
    import diga_amd.util.loss            (package use), or
    sys.path.insert(0, ".../diga_amd"); from util.loss import cross_entropy2d   (drop-in use)
resolves to the same modules.  All compute goes through libdiga_hip.so (include/diga_hip.h).
"""
__all__ = ["build"]

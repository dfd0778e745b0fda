"""Host-side mirror of ``geotransformer.modules`` for the registration hot path (boundary 2)."""

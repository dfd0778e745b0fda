"""Same package layout as the reference's `geotransformer.datasets.registration`, so its import lines keep working."""

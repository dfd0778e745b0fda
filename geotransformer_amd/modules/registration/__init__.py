from .procrustes import WeightedProcrustes, weighted_procrustes

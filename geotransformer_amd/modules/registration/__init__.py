from .matching import get_node_correspondences
from .procrustes import WeightedProcrustes, weighted_procrustes

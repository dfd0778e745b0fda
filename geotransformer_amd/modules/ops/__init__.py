from .grid_subsample import grid_subsample
from .index_select import index_select
from .pairwise_distance import pairwise_distance
from .pointcloud_partition import point_to_node_partition
from .radius_search import radius_search
from .transformation import (apply_rotation, apply_transform, get_rotation_translation_from_transform,
                             get_transform_from_rotation_translation, inverse_transform)

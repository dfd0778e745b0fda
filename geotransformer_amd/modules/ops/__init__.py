from .grid_subsample import grid_subsample
from .radius_search import radius_search

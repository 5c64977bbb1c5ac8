from .nerf_utils import *
from .nerf_ray_query import *

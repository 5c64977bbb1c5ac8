from .neus_utils import *
from .neus_ray_query import *

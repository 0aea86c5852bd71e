"""pine_amd -- MI355X-native PathIntegrator hot path of wicstas/pine."""
from .api import (  # noqa: F401
    Scene, Diffuse, Emissive, Uber, Subsurface, Metal, Glossy, Glass, Node, Position, Normal, UV, Checkerboard, Vec3, lerp,
    node_abs, node_sqr, node_sqrt, node_fract, PointLight, SpotLight, DirectionalLight, Sky, Rect, AABB, OBB, Box, Sphere, Disk, Cone, Mesh, Plane, Line, Cylinder, Triangle,
    Film, Uncharted2, ACES, ThinLenCamera, BlueSampler, SobolSampler, HaltonSampler, PathIntegrator, Plan, PineError,
    mat4, translate, scale, rotate_x, rotate_y, rotate_z, inverse, look_at, film_unpack, packed_offset,
)

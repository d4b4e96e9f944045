"""The reference keeps its curves in geosss/spherical_curve.py; here they live beside the targets that use them
(distributions.py).  This module keeps the reference's import path working:

    from geosss_amd.spherical_curve import SlerpCurve, brownian_curve, constrained_brownian_curve, distance_slerp

Not provided: SphericalSpline (its __call__ needs sphere.map_to_sphere, which the reference does not define) and
SphericalCurve.random_curve (the travelling-salesman ordering of geosss/tsp_solver.py): neither is used by a sampler, a
script or a test of the reference."""
from .distributions import SlerpCurve, brownian_curve, constrained_brownian_curve, distance_slerp

SphericalCurve = SlerpCurve

__all__ = ["SlerpCurve", "SphericalCurve", "brownian_curve", "constrained_brownian_curve", "distance_slerp"]

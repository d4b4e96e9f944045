"""geosss_amd -- many-chain geodesic slice sampling on the sphere for AMD MI355X (gfx950).

Drop-in for the slice-sampler path of microscopic-image-analysis/geosss:

    import geosss_amd as gs
    pdf = gs.MixtureModel([gs.VonMisesFisher(80.0 * mu) for mu in mus])
    samples = gs.ShrinkageSphericalSliceSampler(pdf, init_state, seed).sample(n_samples, burnin)
"""
from . import _lib, diagnostics, io, pointcloud, rand, registration, sphere, spherical_curve
from .diagnostics import IAT, acf, acf_fft, distance, n_eff
from .distributions import (ACG, Bingham, BinghamFisher, CurvedVonMisesFisher, Distribution, MarginalVonMisesFisher, MixtureModel,
                            MultivariateNormal, SlerpCurve, Uniform, VonMisesFisher, brownian_curve, constrained_brownian_curve, distance_slerp,
                            random_bingham)
from .mcmc import (IndependenceSampler, MetropolisHastings, MixtureRWMHIndependenceSampler, RejectionSphericalSliceSampler,
                   ShrinkageSphericalSliceSampler, SphericalHMC, determine_burnin)
from .rand import sample_bingham, sample_bingham_2d, sample_bingham_3d, sample_vMF
from .registration import CoherentPointDrift, GaussianMixtureModel, PointCloud, RotationMatrix, RotationProjection
from .sphere import (cartesian2polar, cartesian2spherical, givens, orthogonal_projection, polar2cartesian, radial_projection,
                     sample_sphere, sample_sphere_device, sample_subsphere, spherical2cartesian, spherical_projection)
from .utils import SamplerLauncher, colors, count_calls, counter, take_time

__all__ = ["Bingham", "BinghamFisher", "CurvedVonMisesFisher", "Distribution", "MixtureModel", "SlerpCurve", "VonMisesFisher",
           "brownian_curve", "random_bingham", "RejectionSphericalSliceSampler", "ShrinkageSphericalSliceSampler",
           "MetropolisHastings", "SphericalHMC", "IndependenceSampler", "MixtureRWMHIndependenceSampler", "determine_burnin", "sample_sphere", "sample_sphere_device", "SamplerLauncher", "count_calls", "counter", "take_time",
           "sphere", "diagnostics", "registration", "rand", "sample_vMF", "sample_bingham", "sample_bingham_2d", "sample_bingham_3d", "CoherentPointDrift", "GaussianMixtureModel", "PointCloud", "RotationProjection", "IAT", "acf", "acf_fft", "distance", "n_eff",
           "io", "pointcloud", "RotationMatrix", "spherical_curve", "constrained_brownian_curve", "distance_slerp", "colors", "ACG", "MarginalVonMisesFisher", "MultivariateNormal", "Uniform", "cartesian2polar", "cartesian2spherical", "givens",
           "orthogonal_projection", "polar2cartesian", "radial_projection", "sample_subsphere", "spherical2cartesian",
           "spherical_projection"]

"""Rigid registration targets on unit quaternions, API-compatible with geosss/registration.py and the parts of
geosss/pointcloud.py they need:

    PointCloud(positions, weights=None)                         pointcloud.py:206-270
    RotationProjection(positions, weights=None)                 pointcloud.py:273-293   (3-D source, 2-D target)
    GaussianMixtureModel(target, source, sigma, k, beta=1)      registration.py:62-118
    CoherentPointDrift(target, source, sigma, k, beta=1, omega=0)   registration.py:186-293
    quat2matrix, matrix2quat                                    pointcloud.py:101-132

`log_prob(rotation)` takes a unit quaternion (x, y, z, w) -- the state of the samplers, S^3 -- or rows of them, or a 3x3
rotation matrix as the reference does, and is evaluated on the GPU (one lane per quaternion; a brute-force scan of the
source cloud with a register-resident list of the k nearest replaces the reference's per-evaluation KD tree).  The slice
samplers, MetropolisHastings and SphericalHMC run on these targets (exact mode); `gradient(rotation)` is the function the
HMC kernel evaluates, callable for rows of quaternions (gsss_gradient).
Translations are not part of the sampled state (the reference's samplers never pass one either); `log_prob(rotation,
translation)` and `gradient(rotation, translation)` take one as the reference's do (tests/test_cpd.py there) and evaluate
on a target moved the other way.
    RotationMatrix(degree=False).create / rotation2d / rotation3d       pointcloud.py:9-98
"""
import contextlib

import numpy as np

from . import _lib
from .distributions import Distribution, _as_f64, counted

__all__ = ["PointCloud", "RotationProjection", "RotationMatrix", "GaussianMixtureModel", "CoherentPointDrift", "quat2matrix",
           "matrix2quat"]


class RotationMatrix:
    """2 x 2 rotation by an angle, or the intrinsic z-y-z rotation R_z(alpha) R_y(beta) R_z(gamma) of three Euler angles
    (pointcloud.py:9-98); `degree=True`: `create` takes degrees.  The last matrix made is kept in `.rot_mat`."""

    def __init__(self, degree=False):
        self.degree = degree
        self.rot_mat = None

    def create(self, angle):
        angle = np.asarray(angle, dtype=np.float64)
        if self.degree:
            angle = np.deg2rad(angle)
        if angle.size == 1:
            return self.rotation2d(float(angle.reshape(())))
        if angle.size == 3:
            return self.rotation3d(angle)
        raise ValueError("Angle vector should be of size 1 or 3")

    @staticmethod
    def _planar(t):
        return np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])

    def rotation2d(self, angle):
        self.rot_mat = self._planar(angle)
        return self.rot_mat

    def rotation3d(self, euler_angles):
        alpha, beta, gamma = (float(a) for a in euler_angles)

        def about_z(t):
            m = np.eye(3)
            m[:2, :2] = self._planar(t)
            return m

        about_y = np.eye(3)
        about_y[np.ix_([0, 2], [0, 2])] = self._planar(beta).T          # (x, z) plane: [[c, s], [-s, c]]
        self.rot_mat = about_z(alpha) @ about_y @ about_z(gamma)
        return self.rot_mat


def quat2matrix(q):
    """Rotation.from_quat(q).as_matrix() (pointcloud.py:101-115): scalar last, the quaternion is normalised."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w]])


def matrix2quat(R):
    """Rotation.from_matrix(R).as_quat() (pointcloud.py:118-132), scalar last."""
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(np.asarray(R, dtype=np.float64)).as_quat()


def _pose_matrix(rotation):
    """A pose given as a unit quaternion (x, y, z, w) or as a matrix -> the matrix."""
    rotation = np.asarray(rotation, dtype=np.float64)
    return quat2matrix(rotation) if rotation.shape[-1] == 4 else rotation


class PointCloud:
    """Parameter carrier of a weighted cloud of n points in the plane or in space: what the device target is packed from
    (`positions` (n, m), `weights` (n,)) plus the pose arithmetic the reference's scripts call on it (pointcloud.py:206-270:
    same constructor, attributes and method names).  `_rows_dropped`: how many trailing rows of a pose matrix do NOT act on the points
    -- none here, the last one for the projected cloud below."""

    _rows_dropped = 0

    def __init__(self, positions, weights=None):
        pts = np.array(positions, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] not in (2, 3):
            raise ValueError(f"a point cloud is an (n, 2) or (n, 3) array, got shape {pts.shape}")
        w = np.ones(len(pts)) if weights is None else np.array(weights, dtype=np.float64)
        if w.shape != (len(pts),):
            raise ValueError(f"one weight per point: expected {(len(pts),)}, got {w.shape}")
        self.positions, self.weights = pts, w

    dim = property(lambda self: self.positions.shape[1], doc="dimension of the space the points live in")
    size = property(lambda self: self.positions.shape[0], doc="number of points")

    @property
    def center_of_mass(self):
        return np.average(self.positions, axis=0, weights=self.weights)

    def transform_positions(self, rotation, translation=None):
        """The points under the pose: R x_i (+ translation), R a unit quaternion or a matrix."""
        R = _pose_matrix(rotation)
        R = R[: len(R) - self._rows_dropped]
        moved = np.einsum("ij,nj->ni", R, self.positions)
        return moved if translation is None else moved + np.asarray(translation, dtype=np.float64)

    def transform(self, rotation, translation=None):
        """Moves the stored points."""
        self.positions = self.transform_positions(rotation, translation)


class RotationProjection(PointCloud):
    """A 3-D cloud seen in parallel projection along z after the rotation (pointcloud.py:273-293): only the first two rows of
    the pose matrix act, the transformed points are 2-D."""

    _rows_dropped = 1


class GaussianMixtureModel(Distribution):
    """Gaussian mixture score of a rigid pose over the k nearest transformed source points of every target point
    (registration.py:62-118)."""

    _outlier = False

    def __init__(self, target, source, sigma=1.0, k=20, *, beta=1.0):
        self.target, self.source = target, source
        self.sigma, self.k, self.beta = float(sigma), int(k), float(beta)
        self.omega = 0.0

    @property
    def d(self):
        return 4

    def _pack(self):
        src, tgt = self.source, self.target
        if src.dim != 3:
            raise ValueError("the source cloud must be 3-D")
        want = 2 if isinstance(src, RotationProjection) else 3
        if tgt.dim != want:
            raise ValueError(f"a {type(src).__name__} source needs a {want}-D target")
        log_volume = float(np.sum(np.log(np.ptp(tgt.positions, 0)))) if self._outlier else 0.0   # registration.py:207-213
        shift = getattr(self, "_shift", None)
        positions = tgt.positions if shift is None else tgt.positions - shift
        extra = dict(source=_as_f64(src.positions), source_w=_as_f64(src.weights), target=_as_f64(positions),
                     target_w=_as_f64(tgt.weights), n_target=tgt.size, target_dim=tgt.dim, k_nn=self.k,
                     outlier=int(self._outlier), sigma=self.sigma, beta=self.beta, omega=self.omega, log_volume=log_volume)
        return _lib.CPD, 4, src.size, 0.0, (None, None, None, None), extra

    @staticmethod
    def _as_quaternions(rotation):
        r = np.asarray(rotation, dtype=np.float64) if not hasattr(rotation, "is_cuda") else rotation
        if not hasattr(r, "is_cuda") and r.shape == (3, 3):
            return matrix2quat(r)
        return r

    @contextlib.contextmanager
    def _shifted(self, translation):
        """A translation of the transformed source is the opposite translation of the target (the score depends on the
        differences only, and the outlier box on the target's extent): evaluate on a target moved by -translation.  The
        sampled state stays the rotation, as in the reference's scripts; a translated model is a second device target."""
        t = None if translation is None else np.asarray(translation, dtype=np.float64)
        if t is not None and t.shape != (self.target.dim,):
            raise ValueError(f"translation must have {self.target.dim} components")
        if t is None or not np.any(t != 0):
            yield
            return
        self._shift = t
        try:
            yield
        finally:
            self._shift = None

    @counted
    def log_prob(self, rotation, translation=None):
        """beta * score of the pose (registration.py:47-53); `rotation`: quaternion (4,), rows (n, 4) or a 3x3 matrix."""
        with self._shifted(translation):
            return self._log_prob_device(self._as_quaternions(rotation))

    def gradient(self, rotation, translation=None):
        """The gradient of the score with respect to the unit quaternion (registration.py:55-60: posterior weights, the 3 x 3
        gradient with respect to R, the quaternion Jacobian of pointcloud.py:135-204) -- what the spherical HMC kernel evaluates;
        `rotation`: quaternion (4,) or rows (n, 4)."""
        with self._shifted(translation):
            return self._gradient_device(self._as_quaternions(rotation))


class CoherentPointDrift(GaussianMixtureModel):
    """GaussianMixtureModel plus a uniform outlier component of weight omega over the target's bounding box
    (registration.py:186-250)."""

    _outlier = True

    def __init__(self, target, source, sigma=1.0, k=20, *, beta=1.0, omega=0.0):
        super().__init__(target, source, sigma, k, beta=beta)
        self.omega = float(omega)

    @counted
    def log_prob(self, rotation, translation=None):
        with self._shifted(translation):
            return self._log_prob_device(self._as_quaternions(rotation))

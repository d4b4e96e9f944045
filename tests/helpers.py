"""Shared helpers of the parity tests: build product-side targets from golden fixtures."""
import numpy as np

COOP_VARIANTS = {8: 16, 9: 32, 10: 64, 11: 128, 12: 256, 13: 512}  # variant id -> max d (gsss_launch.h)
LANE_DIMS = {2: 1, 3: 2, 4: 3, 5: 4, 6: 5, 8: 6, 10: 7}            # d -> variant id


def product_target(z, prefix="target_"):
    import geosss_amd as gs
    kind = str(z[prefix + "kind"])
    if kind == "vmf_mixture":
        return gs.MixtureModel([gs.VonMisesFisher(m) for m in z[prefix + "mu"]], z[prefix + "weights"])
    if kind == "bingham":
        if prefix + "b" in z.files:
            return gs.BinghamFisher(z[prefix + "A"], z[prefix + "b"])
        return gs.Bingham(z[prefix + "A"])
    if kind == "curve_vmf":
        return gs.CurvedVonMisesFisher(gs.SlerpCurve(z[prefix + "knots"]), float(z[prefix + "kappa"]))
    if kind == "cpd":
        tgt = gs.PointCloud(z[prefix + "target"], z[prefix + "target_w"])
        cloud = gs.RotationProjection if z[prefix + "target"].shape[1] == 2 else gs.PointCloud
        src = cloud(z[prefix + "source"], z[prefix + "source_w"])
        if bool(z[prefix + "outlier"]):
            return gs.CoherentPointDrift(tgt, src, float(z[prefix + "sigma"]), int(z[prefix + "k_nn"]),
                                         beta=float(z[prefix + "beta"]), omega=float(z[prefix + "omega"]))
        return gs.GaussianMixtureModel(tgt, src, float(z[prefix + "sigma"]), int(z[prefix + "k_nn"]), beta=float(z[prefix + "beta"]))
    raise ValueError(kind)


def variants_for(d, max_coop=2):
    """Default variant (0) plus up to `max_coop` cooperative layouts that cover d."""
    out = [0]
    coop = [v for v, dmax in sorted(COOP_VARIANTS.items()) if d <= dmax]
    if d not in LANE_DIMS and coop:
        coop = coop[1:]  # the first one IS the default
    return out + coop[:max_coop]


def pad_replay(draws, offsets):
    """Per-step rows of a recorded draw stream (teacher forcing)."""
    n = len(offsets) - 1
    width = int(np.max(np.diff(offsets)))
    out = np.full((n, width), 0.5)
    for i in range(n):
        out[i, : offsets[i + 1] - offsets[i]] = draws[offsets[i]: offsets[i + 1]]
    return out


# shapes the GSSS_MODE_FAST kernels are built for (geosss_amd/csrc/gsss_fast_*.hip)
FAST_BINGHAM = {3, 4, 5, 6, 7, 8, 9, 10}  # lane kernels; 10 < d <= 128 runs the cooperative one


def fast_supported(z, prefix="target_"):
    kind = str(z[prefix + "kind"])
    if kind == "cpd":
        return False
    if kind == "vmf_mixture":
        k, d = z[prefix + "mu"].shape
        return k <= 16 and d <= 256  # lane kernels d <= 10 (component buckets), cooperative fast kernels beyond
    if kind == "bingham":
        d = z[prefix + "A"].shape[0]
        return d in FAST_BINGHAM or 10 < d <= 128  # cooperative fast kernels
    if kind == "curve_vmf":
        k, d = z[prefix + "knots"].shape
        return (3 <= d <= 64 and k <= 16) or (64 < d <= 256 and k <= 17) or (256 < d <= 512 and k <= 10)
    return False


def modes_for(z):
    return ["exact", "fast"] if fast_supported(z) else ["exact"]

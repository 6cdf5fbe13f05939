"""Names of the carrier-peak interpolators `PreshiftDetector` evaluates INSIDE its detection kernel
(`preshift_verdict()` in csrc/detect16k_preshift.hip: float32, like the magnitudes it is given).

The reference passes interpolator FUNCTIONS around (`PreshiftDetector(..., interpolator=gaussian)`,
thrifty/experimental/carrier_interpolators.py); here the four three-point ones are device code and
what travels through the constructor is a selector -- one of the objects below, or its name.  They
compute nothing on the host.  A host interpolator for the DEFAULT detector is any callable
`(fft_mag, peak_idx) -> offset` assigned to `Detector.sync.interpolator` (a slow path between two
engine passes, thrifty_amd/detect.py) -- for the reference's own ones, import them from a Thrifty
installation: `from thrifty.experimental.carrier_interpolators import make_dirichlet`.
"""


class DeviceInterpolator(object):
    """Selector of a device-side interpolator: `name` and the THR_INTERP_* code of thr_create_ex."""

    __slots__ = ("name", "code")

    def __init__(self, name, code):
        self.name, self.code = name, code

    def __call__(self, fft_mag, peak_idx):
        raise NotImplementedError(
            "'%s' runs inside the PreshiftDetector kernel; on the host, assign your own callable "
            "(fft_mag, peak_idx) -> offset to Detector.sync.interpolator" % self.name)

    def __repr__(self):
        return "<device carrier interpolator %r>" % self.name


parabolic = DeviceInterpolator("parabolic", 0)
none = DeviceInterpolator("none", 1)
gaussian = DeviceInterpolator("gaussian", 2)
cosine = DeviceInterpolator("cosine", 3)

INTERPOLATORS = {sel.name: sel for sel in (none, parabolic, gaussian, cosine)}

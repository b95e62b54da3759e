"""Host-side mirror of the reference's nn framework (web_app/components/nn): same class names,
constructor arguments and error behaviour; every tensor op runs in libuniver_hip.so."""
from . import ops  # noqa: F401  (registers the kernel wrappers on CP)
from .gpu import CP, DeviceArray, DeviceScalar  # noqa: F401

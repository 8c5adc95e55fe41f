from .sampler import DPMSolverSampler  # noqa: F401

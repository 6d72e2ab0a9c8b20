"""Fault-injection campaigns (mirror of the reference's ``bnn.faults``, bnn/faults/faults.py)."""
from .faults import CNVFaultTest, FaultTest, LFCFaultTest, NetworkTest  # noqa: F401

__version__ = "0.1"

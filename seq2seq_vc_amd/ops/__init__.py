"""Host-side launchers (kernels.py) and differentiable ops (functional.py) over libs2svc_hip.so."""

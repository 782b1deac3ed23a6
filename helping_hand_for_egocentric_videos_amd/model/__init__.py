"""Host-side mirror of the reference's `model/` package for the hot path (same class names, ctor
arguments, forward signatures and state_dict keys), running on libhh."""

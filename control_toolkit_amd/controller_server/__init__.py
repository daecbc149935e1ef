"""Serving harness: the reference's controller_server loop and its wire format, for driving the
HIP optimizers from a separate process exactly as Controllers/controller_remote.py would."""

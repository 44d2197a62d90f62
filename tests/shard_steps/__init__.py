"""Test-only step modules for the channel-sharded dispatcher (plain torch on CPU tensors): the dispatcher imports any
dotted module name and calls its ``run(data, params)``."""

"""Mirror of the reference's `network` package surface for the per-ray render path
(network/renderer.py, network/render_ops.py, network/dist_decoder.py, network/aggregate_net.py)."""

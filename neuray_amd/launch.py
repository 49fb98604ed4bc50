"""Run one of the reference's scripts, unchanged, on the HIP render path:

    python -m neuray_amd.launch [--render-ops] [--init-nets] [--ft-host] [--render-loop] [--arith x3|f32] <script.py> [script args ...]

e.g. from the reference checkout:  python -m neuray_amd.launch render.py --cfg configs/gen/neuray_gen_depth.yaml ...
The script's directory becomes sys.path[0] (as `python script.py` would make it), `network.renderer` is imported from
there and patched (neuray_amd/integrate.py), then the script runs as __main__.  `--arith x3` sets NEURAY_HIP_ARITH: the MLP contractions of the
inference point kernel on the K = 32 bf16 MFMA with exactly split operands, no yaml edit needed.
"""
import os
import runpy
import sys


def run(script, argv=(), render_ops=False, init_nets=False, ft_host=False, render_loop=False):
    script = os.path.abspath(script)
    root = os.path.dirname(script)
    if root in sys.path:
        sys.path.remove(root)
    sys.path.insert(0, root)
    from . import integrate
    integrate.patch_reference(render_ops=render_ops, init_nets=init_nets, ft_host=ft_host, render_loop=render_loop)
    old = sys.argv
    sys.argv = [script] + list(argv)
    try:
        return runpy.run_path(script, run_name='__main__')
    finally:
        sys.argv = old


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    opts = {'render_ops': False, 'init_nets': False, 'ft_host': False, 'render_loop': False}
    while argv and argv[0] in ('--render-ops', '--init-nets', '--ft-host', '--render-loop', '--arith'):
        flag = argv.pop(0)
        if flag == '--arith':
            if not argv or argv[0] not in ('x3', 'f32'):
                raise SystemExit("--arith takes x3 or f32")
            os.environ['NEURAY_HIP_ARITH'] = argv.pop(0)
        else:
            opts[flag[2:].replace('-', '_')] = True
    if not argv:
        raise SystemExit(__doc__)
    run(argv[0], argv[1:], **opts)


if __name__ == '__main__':
    main()

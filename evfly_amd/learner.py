"""`argparsing` -- the configuration surface `evfly_ros/run.py:61` consumes.

Mirror of learner/learner.py:1167-1272 without the configargparse dependency (not installed in
the build image): the same option names, types, `nargs`, defaults and `action`s, the same
config-file syntax (`key = value`, lists as `[a, b]`, booleans as `True`/`False`,
`checkpoint_path` appended per list entry, unknown keys ignored via parse_known_args), and
command-line arguments overriding the file. The training side of learner.py (Learner) is out of
scope.
"""
import argparse
import getpass
import os
import sys

try:
    uname = getpass.getuser()
except Exception:  # pragma: no cover
    uname = "user"


def _spec():
    """(name, kwargs) for every option of learner/learner.py:1178-1265, in its order."""
    S, I, F = str, int, float
    home = f'/home/{uname}/evfly_ws/src/evfly'
    flag = dict(action='store_true')
    return [
        ('basedir', dict(type=S, default=home)), ('logdir', dict(type=S, default='learner/logs')),
        ('datadir', dict(type=S, default=home)),
        ('ws_suffix', dict(type=S, default='')), ('model_type', dict(nargs='+', type=S, default='LSTMNet')),
        ('velpred', dict(type=I, default=0)), ('dataset', dict(nargs='+', type=S, default=None)),
        ('use_h5', flag), ('short', dict(type=I, default=0)), ('val_split', dict(type=F, default=0.2)),
        ('seed', dict(type=I, default=None)), ('batch_size', dict(type=I, default=0)),
        ('device', dict(type=S, default='cuda')), ('load_trainval', flag),
        ('checkpoint_path', dict(action='append')), ('lr', dict(type=F, default=1e-4)),
        ('N_eps', dict(type=I, default=100)), ('lr_warmup_epochs', dict(type=I, default=5)), ('lr_decay', flag),
        ('save_model_freq', dict(type=I, default=25)), ('val_freq', dict(type=I, default=10)),
        ('optional_loss_param', dict(nargs='+', type=F, default=None)),
        ('num_recurrent', dict(nargs='+', type=I, default=0)), ('events', dict(type=S, default='')),
        ('keep_collisions', flag), ('do_transform', flag), ('eval_tools_freq', dict(type=I, default=0)),
        ('eval_tools_on_best', flag), ('print_trainprogress_freq', dict(type=I, default=1)),
        ('num_out_channels', dict(type=I, default=1)), ('num_in_channels', dict(type=I, default=2)),
        ('resize_input', dict(nargs='+', type=I, default=None)), ('loss_weights', dict(nargs='+', type=F, default=None)),
        ('split_method', dict(type=S, default='train-val')), ('num_outputs', dict(type=I, default=2)),
        ('rescale_depth', dict(type=F, default=0.0)), ('rescale_evs', dict(type=F, default=0.0)),
        ('domain_randomization', dict(type=F, default=0.0)), ('bev', dict(type=I, default=0)),
        ('skip_type', dict(type=S, default='crop')), ('combine_checkpoints', flag),
        ('data_augmentation', dict(type=F, default=0.0)), ('evs_min_cutoff', dict(type=F, default=0.0)),
        # encoder / decoder / fc blocks of the velpred heads
        ('enc_num_layers', dict(type=I, default=2)), ('enc_kernel_sizes', dict(nargs='+', type=I, default=[5, 5])),
        ('enc_kernel_strides', dict(nargs='+', type=I, default=[2, 2])),
        ('enc_out_channels', dict(nargs='+', type=I, default=[16, 64])),
        ('enc_activations', dict(nargs='+', type=S, default=['relu', 'relu'])), ('enc_pool_type', dict(type=S, default='max')),
        ('enc_invert_pool_inputs', flag), ('enc_pool_kernels', dict(nargs='+', type=I, default=[2, 2])),
        ('enc_pool_strides', dict(nargs='+', type=I, default=[2, 2])), ('enc_conv_function', dict(type=S, default='conv2d')),
        ('dec_num_layers', dict(type=I, default=2)), ('dec_kernel_sizes', dict(nargs='+', type=I, default=[5, 5])),
        ('dec_kernel_strides', dict(nargs='+', type=I, default=[2, 2])),
        ('dec_out_channels', dict(nargs='+', type=I, default=[64, 16])),
        ('dec_activations', dict(nargs='+', type=S, default=['relu', 'sigmoid'])), ('dec_pool_type', dict(type=S, default='max')),
        ('dec_pool_kernels', dict(nargs='+', type=I, default=[2, 2])), ('dec_pool_strides', dict(nargs='+', type=I, default=[2, 2])),
        ('dec_conv_function', dict(type=S, default='upconv2d')),
        ('fc_num_layers', dict(type=I, default=3)), ('fc_layer_sizes', dict(nargs='+', type=I, default=[128, 32, 1])),
        ('fc_activations', dict(nargs='+', type=S, default=['leaky_relu', 'leaky_relu', 'tanh'])),
        ('fc_dropout_p', dict(type=F, default=0.1)),
        # deployment compatibility flags
        ('align_evframe', flag), ('vision_based', flag), ('ppo_path', dict(default=None)),
        ('model_path', dict(type=S, default=None)), ('keyboard', flag), ('planner', flag),
    ]


def _split_list(text):
    text = text.strip()
    if text.startswith('[') and text.endswith(']'):
        return [t.strip().strip('\'"') for t in text[1:-1].split(',') if t.strip()]
    return None


def _config_to_argv(path, spec):
    """Translate a `key = value` config file into argv tokens (configargparse's file semantics)."""
    kinds = {name: kw for name, kw in spec}
    argv = []
    with open(path) as f:
        for raw in f:
            line = raw.split('#', 1)[0].strip()
            if not line or line.startswith(';'):
                continue
            if '=' in line:
                key, val = (s.strip() for s in line.split('=', 1))
            elif ':' in line:
                key, val = (s.strip() for s in line.split(':', 1))
            else:
                key, val = line, 'True'
            key = key.lstrip('-')
            kw = kinds.get(key)
            if kw is None:
                continue                                   # unknown keys are ignored (parse_known_args, :1268)
            items = _split_list(val)
            if kw.get('action') == 'store_true':
                if val.lower() in ('true', 'yes', '1'):
                    argv.append('--' + key)
            elif kw.get('action') == 'append':
                for it in (items if items is not None else [val]):
                    argv += ['--' + key, it]
            elif 'nargs' in kw:
                argv += ['--' + key] + (items if items is not None else val.split())
            else:
                argv += ['--' + key, val]
    return argv


def argparsing(filename=None, argv=None):
    """learner/learner.py:1167-1272: returns the argparse.Namespace the models are built from."""
    if filename is not None:
        default_config_files = [filename]
    else:
        default_config_files = [f'/home/{uname}/evfly_ws/src/evfly/learner/configs/config.txt']   # :1172
    spec = _spec()
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', default=None, help='config file relative path')
    for name, kw in spec:
        parser.add_argument('--' + name, **kw)
    cli = list(sys.argv[1:] if argv is None else argv)
    pre, _ = parser.parse_known_args(cli)
    files = [pre.config] if pre.config else [p for p in default_config_files if os.path.exists(p)]
    file_argv = []
    for p in files:
        file_argv += _config_to_argv(p, spec)
    args, unknown = parser.parse_known_args(file_argv + cli)       # command line wins over the file
    if args.config is None and files:
        args.config = files[0]
    print(f'[CONFIGARGPARSE] Parsing args from config file {args.config}')
    return args

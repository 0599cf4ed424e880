"""`learner.argparsing` (learner/learner.py:1167-1272) without configargparse: file syntax, types, overrides."""
import textwrap

from _util import GOLDEN  # noqa: F401
from evfly_amd.learner import argparsing

CFG = textwrap.dedent("""\
    device = cuda
    dataset = [real_forest-a]
    evs_min_cutoff = 0.15
    align_evframe = True
    use_h5 = True
    do_transform = False
    combine_checkpoints = True
    checkpoint_path = [/models/D_theta.pth, /models/V_phi.pth]
    lr = 1e-4
    loss_weights = [10.0, 1.0]
    model_type = [OrigUNet, VITFLY_ViTLSTM]
    skip_type = interp
    velpred = 0
    num_in_channels = 2
    bev = 2
    num_recurrent = [1, 0]
    resize_input = [260, 346]
    enc_kernel_sizes = [5, 3]
    enc_activations = [relu, relu]
    fc_layer_sizes = [1024, 128, 16, 1]
    some_unknown_key = 42
    """)


def test_config_file(tmp_path):
    p = tmp_path / "config.txt"
    p.write_text(CFG)
    a = argparsing(filename=str(p), argv=[])
    # the hyper-parameters evfly_ros/run.py:106-137 builds the deployed model from
    assert a.model_type == ["OrigUNet", "VITFLY_ViTLSTM"] and a.skip_type == "interp" and a.velpred == 0
    assert a.num_in_channels == 2 and a.bev == 2 and a.num_recurrent == [1, 0] and a.resize_input == [260, 346]
    assert a.evs_min_cutoff == 0.15 and a.device == "cuda" and a.dataset == ["real_forest-a"]
    assert a.align_evframe is True and a.use_h5 is True and a.do_transform is False and a.combine_checkpoints is True
    assert a.checkpoint_path == ["/models/D_theta.pth", "/models/V_phi.pth"]          # action='append'
    assert a.loss_weights == [10.0, 1.0] and a.lr == 1e-4
    assert a.enc_kernel_sizes == [5, 3] and a.enc_activations == ["relu", "relu"] and a.fc_layer_sizes == [1024, 128, 16, 1]
    # defaults of options the file does not mention (learner.py:1178-1265)
    assert a.num_out_channels == 1 and a.dec_conv_function == "upconv2d" and a.fc_dropout_p == 0.1 and a.short == 0
    assert a.val_split == 0.2 and a.N_eps == 100 and a.keyboard is False and a.model_path is None
    assert not hasattr(a, "some_unknown_key")


def test_cli_overrides_file(tmp_path):
    p = tmp_path / "config.txt"
    p.write_text(CFG)
    a = argparsing(filename=str(p), argv=["--bev", "0", "--skip_type", "crop", "--num_recurrent", "0", "0", "--junk", "1"])
    assert a.bev == 0 and a.skip_type == "crop" and a.num_recurrent == [0, 0]


def test_defaults_without_file():
    a = argparsing(filename="/nonexistent/config.txt", argv=[])
    assert a.model_type == "LSTMNet" and a.num_recurrent == 0 and a.checkpoint_path is None and a.bev == 0

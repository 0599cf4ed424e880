"""The reference's own import lines (evfly_ros/run.py:23-30) resolve to evfly_amd through evfly_amd/compat."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_py_import_surface():
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.join(REPO, 'evfly_amd', 'compat')!r})
        from ev_utils import simple_evim                                    # run.py:24, verbatim
        from ev_utils import form_eventframe
        from learner import argparsing
        from learner_models import *
        import vitfly_models
        from ConvLSTM_pytorch.convlstm import ConvLSTM
        from calibration_tools.rectify_bag import Aligner                   # run.py:25
        m = OrigUNet_w_VITFLY_ViTLSTM(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0],
                                      input_shape=[1, 1, 260, 346], velpred=0, enc_params={{}}, dec_params={{}}, fc_params={{}},
                                      form_BEV=2, evs_min_cutoff=0.15, skip_type='interp', is_deployment=False)
        assert sum(p.numel() for p in m.parameters()) == 13420336          # BASELINE.md: composite parameter count
        assert isinstance(m.vitfly_vitlstm, vitfly_models.LSTMNetVIT) and callable(form_eventframe) and callable(argparsing) and callable(Aligner) and callable(simple_evim)
        print('ok')
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]

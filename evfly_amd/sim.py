"""Non-ROS mirror of the simulator pilot's vision path (envtest/ros/run_competition.py).

`AgilePilotVision` keeps the data flow and attribute names of `AgilePilotNode` for the part that touches the
model, without rospy / cv_bridge / the flightmare messages:

  im_callback(gray float32 image)            :962-999   store im / prev_im, estimate events from their log difference
  compute_events(neg_thresh, pos_thresh)     :603-635   -> evfly_difflog_events
  compute_command_vision_based()             :466-584   resize, q97 scaling + clamp, stateful model call, hidden-state
                                                         hand-off per model type, velocity post-scale, acceleration ramp

Model construction follows :211-262 (note `evs_min_cutoff=0.0` there, unlike run.py's 0.15) and the checkpoint
combination of :440-459. SURVEY.md §8f row N4. All arithmetic is native (include/evfly_hip.h); there is no CPU path.
"""
import numpy as np
import torch

from . import learner_models, voxelizer


def combine_state_dicts(state_dicts, model_names=None):
    """run_competition.py:440-459: first state dict wins; keys optionally prefixed `<model_name>.`."""
    combined = {}
    for i, sd in enumerate(state_dicts):
        for key, value in sd.items():
            if model_names is not None:
                key = f"{model_names[i]}.{key}"
            if key not in combined:
                combined[key] = value
    return combined


def build_model(args, enc_params=None, dec_params=None, fc_params=None, logger=None):
    """run_competition.py:211-262 for the two model types this path serves."""
    mt = args.model_type
    kw = dict(num_in_channels=args.num_in_channels, num_out_channels=args.num_out_channels,
              num_recurrent=args.num_recurrent, input_shape=[1, 1, args.resize_input[0], args.resize_input[1]],
              velpred=args.velpred, enc_params=enc_params, fc_params=fc_params, form_BEV=args.bev, evs_min_cutoff=0.0,
              skip_type=args.skip_type, logger=logger)
    if mt == 'OrigUNet' or (isinstance(mt, list) and len(mt) == 1 and mt[0] == 'OrigUNet'):
        return learner_models.OrigUNet(**kw)
    if mt == 'OrigUNet_w_VITFLY_VitLSTM' or (isinstance(mt, list) and mt[0] == 'OrigUNet' and mt[1] == 'VITFLY_ViTLSTM'):
        return learner_models.OrigUNet_w_VITFLY_ViTLSTM(dec_params=dec_params, is_deployment=False, **kw)
    raise ValueError(f'[RUN_COMPETITION] Invalid self.args.model_type {mt}.')


class AgilePilotVision:
    def __init__(self, model, resize_input=(260, 346), image_hw=(260, 346), desiredVel=4.0, num_recurrent=(1, 0),
                 device="cuda"):
        self.model = model.to(device).float().eval()
        self.device = device
        self.resize_input = tuple(resize_input)
        self.image_h, self.image_w = image_hw
        self.desiredVel = desiredVel
        self.num_recurrent = list(num_recurrent)
        self.do_events = True                                             # :74
        self.im = np.zeros((self.image_h, self.image_w), dtype=np.float32)      # :338-341
        self.prev_im = np.zeros((self.image_h, self.image_w), dtype=np.float32)
        self.events = torch.zeros(self.image_h, self.image_w, device=device)    # :196
        self.model_hidden_state = [None]                                  # :318
        self.extras = None
        self.im_ctr = 0
        self._composite = isinstance(self.model, learner_models.OrigUNet_w_VITFLY_ViTLSTM)

    # ------------------------------------------------------------------ :962-999
    def im_callback(self, im):
        """im: (H, W) gray image, uint8 or float32/255 (the reference converts bgr8 -> gray -> float32/255)."""
        im = np.asarray(im)
        if im.dtype == np.uint8:
            im = im.astype(np.float32) / 255.0
        self.im_ctr += 1
        self.prev_im = self.im
        self.im = im.astype(np.float32, copy=False)
        if self.do_events:
            self.compute_events()

    def compute_events(self, neg_thresh=0.2, pos_thresh=0.2):
        """:603-635. The frame stays on the device."""
        if self.im is None or self.prev_im is None:
            self.events = torch.zeros(self.image_h, self.image_w, device=self.device)
            return
        self.events = voxelizer.difflog_events(self.im, self.prev_im, pos_thresh, neg_thresh)[0]

    # ------------------------------------------------------------------ :466-584
    def reset_hidden_state(self):
        if self._composite:
            self.model_hidden_state = ((None, None), None)               # :515
        else:
            self.model_hidden_state = [[None, None]]                     # :510

    def compute_command_vision_based(self, pos_x=10.0):
        """Returns the LINVEL command velocity (3,) float64 for the current `self.events`. `pos_x` is
        `self.state.pos[0]` (hidden-state reset below 0.5 m :503, acceleration ramp below 2 m :567-571)."""
        im = self.events
        if tuple(im.shape[-2:]) != self.resize_input:                     # :487-488
            im = voxelizer.resize_bilinear(im, self.resize_input)[0]
        if sum(self.num_recurrent) > 0 and (pos_x < 0.5 or self.model_hidden_state is None
                                            or self.model_hidden_state == [None]):
            self.reset_hidden_state()
        # :491-493 + :537: q97 of |im|, clamp(im / q, -1, 1) -- one fused native pass
        frame = voxelizer.condition_frames(im[None], out_hw=self.resize_input, quantile=0.97)
        desvel = torch.tensor(self.desiredVel, device=frame.device).view(1, 1).float()
        with torch.no_grad():
            out = self.model([frame, desvel, *self.model_hidden_state])
        x, self.extras = out
        if self._composite:
            self.model_hidden_state = self.extras[2]                      # :547-549
        else:
            self.model_hidden_state = [self.extras[2]]                    # :555-557
        x = x.squeeze().detach().cpu().numpy().astype(np.float64)
        velocity = x * self.desiredVel                                    # :577
        min_xvel_cmd, hardcoded_ctl_threshold = 1.0, 2.0                  # :582-585
        if pos_x < hardcoded_ctl_threshold:
            velocity[0] = max(min_xvel_cmd, (pos_x / hardcoded_ctl_threshold) * self.desiredVel)
        return velocity

"""InterfacePhysics: the reference's training-interface surface for the physics-informed step
(interface/interface_physics.py:32-332 and the step bodies :443-515 / :990-1065).

Hot methods (`place_one_batch`, `data_loss`, `training_step`) run the fused HIP point path.  The six
`*_equation` methods and `gradient` keep the reference's signatures and formulas as plain torch expressions, for
callers that hold autograd-connected fields of their own.  File I/O, visualisation and the epoch loops of the
reference are out of scope.
"""
import os
import shutil

import torch
import torch.nn as nn

from ..losses.builder import builder_loss
from ..model.physics_net import PhysicsNet
from ..point_path import LOSS_ORDER, OBS_ORDER, PointConfig, pde_losses, smooth_l1_data_loss
from ..utils.position_encoding import SineCosPE
from .. import _lib as L


class InterfacePhysics(nn.Module):
    def __init__(self, meta_cfg: dict, net_cfg: dict, obs_norm_cfg: dict, variable_cfg: dict, train_cfg: dict, test_cfg=None,
                 inference_cfg: dict = None, precision='bf16x2', **kwargs):
        super().__init__()
        self.net_cfg, self.obs_norm_cfg, self.variable_cfg = net_cfg, obs_norm_cfg, variable_cfg
        self.train_cfg, self.test_cfg, self.inference_cfg = train_cfg, test_cfg, inference_cfg
        self.physics_net = PhysicsNet(meta_cfg, net_cfg)
        self.pe = SineCosPE(3, include_input=False)
        img_size = self.train_cfg['img_size']
        if isinstance(img_size, (int, float)):
            self.lat_size, self.lon_size = img_size, img_size
        elif isinstance(img_size, (list, tuple)) and len(img_size) == 2:
            self.lat_size, self.lon_size = img_size
        else:
            raise NotImplementedError
        # attributes the reference only sets inside its training loops (:339-342, :412-418, :438)
        self.dx = float(train_cfg.get('dx', 27000.0))
        self.dy = float(train_cfg.get('dy', 27000.0))
        self.dt = 3600.0 * float(train_cfg.get('lable_time_step', 1))
        td = train_cfg.get('train_data', {})
        self.pred_t_span = float(td.get('input_time_step', 6)) * float(td.get('input_time_step_nums', 4)) * 3600.0
        self.with_clip = True
        self.precision = L.PREC_NAMES[precision]
        self._cfg_cache = None

    # ------------------------------------------------------------------ configuration of the HIP path
    def point_config(self, loss_factor=None, criterion=None) -> PointConfig:
        crit_kind, crit_beta, crit_sum = self._check_pde_criterion(criterion if criterion is not None else self.train_cfg['losses'].get('pde_loss', {'name': 'MSELoss'}))
        # inverse_norm (:232-262) per variable as (out * std + mean) [squared and shifted] + optional clip: mean_norm v * nf[1] + nf[0]; min_max
        # v * (nf[1] - nf[0]) + nf[0], with a third factor (v * (nf[1] - nf[0]) + nf[0]) ** 2 + nf[2] (:244-247); use_norm False: the identity and no
        # clip (the reference clips inside its `if use_norm`: bounds of +-FLT_MAX never bind)
        mean, std, clipv, sq = [], [], [], []
        for k in OBS_ORDER:
            c = self.obs_norm_cfg[k]
            if not c.get('use_norm', True):
                mean.append(0.0), std.append(1.0), clipv.append(False), sq.append(None)
                continue
            nf = c['norm_factor']
            if c.get('norm_type', 'mean_norm').lower() == 'min_max':
                mean.append(float(nf[0])), std.append(float(nf[1]) - float(nf[0]))
                sq.append(None if len(nf) == 2 else float(nf[2]))
            else:
                mean.append(float(nf[0])), std.append(float(nf[1])), sq.append(None)
            clipv.append(True)
        lf = loss_factor or self.train_cfg['losses']['loss_factor']
        big = 3.4028234663852886e38
        bound = lambda k, i: (float(self.obs_norm_cfg[k]['bound'][i]) if ('bound' in self.obs_norm_cfg[k] and self.obs_norm_cfg[k].get('use_norm', True))
                              else (-big, big)[i])
        bounds = tuple((bound(k, 0), bound(k, 1)) for k in OBS_ORDER)        # part of the key: a changed clip bound alone must rebuild the configuration
        key = (self.dx, self.dy, self.lon_size, self.lat_size, self.pred_t_span, bool(self.with_clip), self.precision,
               tuple(float(lf[k]) for k in LOSS_ORDER), tuple(mean), tuple(std), tuple(clipv), crit_kind, crit_beta, tuple(sq), crit_sum, bounds)
        if self._cfg_cache is None or self._cfg_cache[0] != key:
            cfg = PointConfig(dx=self.dx, dy=self.dy, lon_size=self.lon_size, lat_size=self.lat_size, pred_t_span=self.pred_t_span,
                              mean=tuple(mean), std=tuple(std),
                              clip_lo=tuple(bound(k, 0) for k in OBS_ORDER), clip_hi=tuple(bound(k, 1) for k in OBS_ORDER),
                              with_clip=bool(self.with_clip), clip_vars=tuple(clipv), factors=key[7], prec=self.precision,
                              criterion=crit_kind, beta=crit_beta, sq_add=tuple(sq), reduce_sum=crit_sum)
            self._cfg_cache = (key, cfg)
        self.physics_net.point_cfg = self._cfg_cache[1]
        return self._cfg_cache[1]

    # ------------------------------------------------------------------ checkpoints (:53-88)
    def save_model(self, checkpoint_path, epoch, global_step, prefix='physics', **kwargs):
        checkpoint_file = os.path.join(checkpoint_path, '%s_%d.pth' % (prefix, epoch))
        state_dict = {'model': self.physics_net.state_dict(), 'epoch': epoch, 'gobal_step': global_step}
        state_dict.update(kwargs)
        torch.save(state_dict, checkpoint_file)
        shutil.copy(checkpoint_file, os.path.join(checkpoint_path, '%s_latest.pth' % prefix))

    def load_model(self, checkpoint_path, current_epoch=None, prefix='downscale', map_location='cpu'):
        if os.path.isfile(checkpoint_path):
            model_file = checkpoint_path
        elif current_epoch is None:
            model_file = os.path.join(checkpoint_path, '%s_latest.pth' % prefix)
        else:
            model_file = os.path.join(checkpoint_path, '%s_%d.pth' % (prefix, current_epoch))
        if not os.path.exists(model_file):
            print('warning:%s does not exist!' % model_file)
            return None, 0, 0
        state_dict = torch.load(model_file, map_location=map_location)
        glob_step = state_dict.pop('gobal_step', 0)
        epoch = state_dict.pop('epoch', 0)
        # checkpoints written under DDP carry a 'module.' prefix (:1397 via :56); accept both
        if 'model' in state_dict and any(k.startswith('module.') for k in state_dict['model']):
            state_dict['model'] = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in state_dict['model'].items()}
        return state_dict, epoch + 1, glob_step

    # ------------------------------------------------------------------ generic torch expressions (reference formulas)
    def gradient(self, y, x):
        return torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), create_graph=True, only_inputs=True, allow_unused=False)[0]

    def montion_equation_u(self, x, y, t, u, v, p, rio, f, loss, factor=1e-6):
        lhs = self.gradient(u, t) + u * self.gradient(u, x) + v * self.gradient(u, y) + self.gradient(p, x) / rio
        return loss(lhs, f * v).float() * factor

    def montion_equation_v(self, x, y, t, u, v, p, rio, f, loss, factor=1e-6):
        lhs = self.gradient(v, t) + u * self.gradient(v, x) + v * self.gradient(v, y) + self.gradient(p, y) / rio
        return loss(lhs, -f * u).float() * factor

    def continuous_equation(self, x, y, t, u, v, rio, loss, factor=1e-6):
        lhs = (self.gradient(rio, t) + u * self.gradient(rio, x) + v * self.gradient(rio, y)
               + rio * self.gradient(u, x) + rio * self.gradient(v, y))
        return loss(lhs, torch.zeros_like(lhs).float()).float() * factor

    def _advect(self, a, x, y, t, u, v):
        return self.gradient(a, t) + u * self.gradient(a, x) + v * self.gradient(a, y)

    def energy_equation(self, x, y, t, u, v, p, T, rio, q, loss, factor=1e-6, c_p=1005, L=2.5e6):
        lhs = c_p * self._advect(T, x, y, t, u, v) + (-self._advect(p, x, y, t, u, v) / (rio + 1e-6)) + L * self._advect(q, x, y, t, u, v)
        return loss(lhs, torch.zeros_like(lhs).float()).float() * factor

    def get_qs(self, p, T):
        tc = T - 273.15
        e_s = 6.112 * torch.exp(17.67 * tc / (tc + 243.5)) * 100
        return 0.622 * e_s / (p - 0.378 * e_s)

    def vapor_equation(self, x, y, t, u, v, p, T, q, loss, factor=1e-5, c_p=1005, L=2.5e6, R_v=461.5, R_d=287):
        omega = self._advect(p, x, y, t, u, v)
        q_adv = self._advect(q, x, y, t, u, v)
        q_s = torch.maximum(self.get_qs(p, T).detach(), torch.full_like(p, 1e-6))
        delta = ((omega < 0) & (q >= q_s)).to(p.dtype).detach()
        R = (1 + 0.608 * q) * R_d
        Fv = (((L * R - c_p * R_v * T) / (c_p * R_v + T * T + L * L * q_s)) * q_s * T).detach()
        lhs = -omega * delta * Fv / (p + 1e-6) + q_adv
        return loss(lhs, torch.zeros_like(lhs).float()).float() * factor

    def gas_equation(self, p, T, rio, q, loss, factor=1e-5, R_d=287):
        return loss(p, rio * (1 + 0.608 * q) * R_d * T).float() * factor

    def inverse_norm(self, u, v, P, T, q, rio, obs_norm_cfg, with_clip=False):
        """:232-262 -- like the reference, clipping follows self.with_clip and never applies to u, v."""
        out = []
        for k, (val, name) in enumerate(zip((u, v, P, T, q, rio), OBS_ORDER)):
            c = obs_norm_cfg[name]
            if c['use_norm']:
                nf = c['norm_factor']
                if c['norm_type'].lower() == 'min_max':                 # :242-249 (unused by the shipped config)
                    val = val * (nf[1] - nf[0]) + nf[0]
                    if len(nf) != 2:
                        val = val ** 2 + nf[2]
                else:
                    val = val * nf[1] + nf[0]
                if k >= 2 and self.with_clip:
                    val = torch.clip(val, c['bound'][0], c['bound'][1])
            out.append(val)
        return tuple(out)

    def calc_rio(self, p, T, q, R_d=287):
        return ((1 + 0.608 * q) * R_d * T / p).detach()

    def encoding_coord(self, x, y, t, pred_t_span):
        x = x / self.dx / (self.lon_size - 1)
        y = y / self.dy / (self.lat_size - 1)
        t = t / pred_t_span
        pts = torch.stack([x, y, t], dim=1) if x.dim() == 1 else torch.cat([x, y, t], dim=1)
        return self.pe(pts)

    # ------------------------------------------------------------------ fused HIP path
    def _check_pde_criterion(self, criterion):
        """The PDE criterion (`builder_loss(**train_cfg.losses.pde_loss)`, :384; every equation calls it as loss(residual, 0), :104 ... :179) as the
        fused residual kernel's (kind, beta, reduce_sum).  What the reference's losses/builder.py can build for it is implemented: MSELoss (cfg:137),
        L1Loss -- reduction "mean" or "sum" --, WeightSmoothL1Loss(beta); given as a module or as the config's dict."""
        from .. import _lib as L
        from ..losses import WeightSmoothL1Loss
        red = {'mean': False, 'sum': True}
        if isinstance(criterion, dict):
            name = criterion.get('name', 'MSELoss')
            extra = set(criterion) - {'name'}
            if name in ('MSELoss', 'L1Loss') and extra <= {'reduction'} and criterion.get('reduction', 'mean') in red:
                return (L.CRIT_MSE if name == 'MSELoss' else L.CRIT_L1), 0.0, red[criterion.get('reduction', 'mean')]
            if name == 'WeightSmoothL1Loss' and extra <= {'beta'} and float(criterion.get('beta', 0.1)) > 0.0:
                return L.CRIT_SMOOTH_L1, float(criterion.get('beta', 0.1)), False
        elif isinstance(criterion, (nn.MSELoss, nn.L1Loss)) and criterion.reduction in red:
            return (L.CRIT_MSE if isinstance(criterion, nn.MSELoss) else L.CRIT_L1), 0.0, red[criterion.reduction]
        elif isinstance(criterion, WeightSmoothL1Loss) and criterion.beta > 0:
            return L.CRIT_SMOOTH_L1, float(criterion.beta), False
        elif isinstance(criterion, nn.SmoothL1Loss) and criterion.reduction in red and criterion.beta > 0:
            return L.CRIT_SMOOTH_L1, float(criterion.beta), red[criterion.reduction]
        raise NotImplementedError('the fused residual kernel implements the PDE criteria of the reference\'s loss builder -- nn.MSELoss (cfg:137), nn.L1Loss, '
                                  'WeightSmoothL1Loss(beta > 0), reduction "mean" or "sum"; got %r' % (criterion,))

    def pde_loss_terms(self, x, y, t, f, field_data, input_data, forecast_h, loss_factor=None, use_cache=False, with_total=False, criterion=None):
        """The six scaled residual losses as a [6] tensor (motion_u, motion_v, continuous, energy, vapor, gas)."""
        cfg = self.point_config(loss_factor, criterion)
        heads, evec, statics = self.physics_net.field_weights(field_data, forecast_h, use_cache=use_cache)
        return pde_losses(cfg, x, y, t, f, input_data, heads, evec, statics, with_total=with_total)

    def place_one_batch(self, x, y, t, f, field_data, input_data, forecast_h, criterion, loss_factor, global_step, local_rank, device,
                        summary=None, prefix='inter', log_step=100, use_cache=False):
        """:271-320.  Same arguments and return value; the fields, the Jacobian, the residuals and their backward run in HIP."""
        f, x, y, t = f.to(device), x.to(device), y.to(device), t.to(device)
        # train_loss = mu + mv + en + co + va + ga in the reference's order of additions (:301), formed inside the residual kernel
        terms, train_loss = self.pde_loss_terms(x, y, t, f, field_data, input_data, forecast_h, loss_factor, use_cache=use_cache,
                                                with_total=True, criterion=criterion)
        if summary is not None and global_step % log_step == 1 and local_rank == 0:
            names = ('montion_u_loss', 'montion_v_loss', 'continous_loss', 'energy_loss', 'vapor_loss', 'gas_loss')
            vals = terms.detach().cpu().tolist()
            summary.add_scalar('%s/total_loss' % prefix, float(train_loss.detach()), global_step)
            for n_, v_ in zip(names, vals):
                summary.add_scalar('%s/%s' % (prefix, n_), v_, global_step)
            print('%s:' % prefix + ','.join('%s:%f' % (n_, v_) for n_, v_ in zip(names, vals)))
        return train_loss.float()

    def place_lead_batch(self, x, y, t, f, field_data, input_data, forecast_h, criterion, loss_factor, reduction='mean'):
        """BASELINE configs[2]: B field samples (e.g. the 61 forecast leads, physics_dataset.py:173-174) in ONE step.  x, y, t, f: [B, N];
        field_data [B, 159, 2405]; input_data [B, N, 6]; forecast_h [B, 1, 1].  Equals place_one_batch applied to every sample
        (:271-320), the B totals averaged ('mean') or added ('sum'); the encoder runs once over all B samples and the point kernels
        field after field.  Returns (loss, terms [B, 6])."""
        from ..point_path import pde_losses_batch
        cfg = self.point_config(loss_factor, criterion)
        heads, evec, statics = self.physics_net.field_weights(field_data, forecast_h)
        B = field_data.shape[0]
        heads, evec = heads.reshape(B, 256, -1), evec.reshape(B, 6, 256)
        terms, totals = pde_losses_batch(cfg, x, y, t, f, input_data, heads, evec, statics)
        loss = totals.sum() if reduction == 'sum' else totals.mean()
        return loss.float(), terms

    def data_loss(self, x, y, t, field_data, input_data, labels, forecast_h, margin_factor=None, beta=0.1, use_cache=False):
        """Data ("margin") loss of the step body (:464-474): mean SmoothL1(beta) over [N,6] times margin_factor."""
        self.point_config()
        if margin_factor is None:
            margin_factor = self.train_cfg['losses']['loss_factor']['margin_factor']
        fields = self.physics_net.forward_xyt(field_data, x, y, t, input_data, forecast_h, use_cache=use_cache)
        return smooth_l1_data_loss(torch.cat(fields, dim=1), labels, beta=beta, factor=1.0).float() * margin_factor

    @torch.no_grad()
    def predict_grid(self, field_data, x, y, t, input_data, forecast_h, with_clip=False, use_cache=False):
        """Full-grid evaluation of the visualisation branch (:536-591): the six fields at all lon*lat nodes, given in the
        reference's node order (x outer, y inner; e.g. CollocationSampler.full_grid), de-normalised (the reference switches
        the clip off there, :533) and scattered into maps [6, lat, lon] (u, v, P, T, q, rho) on the device."""
        import ctypes
        from .. import _lib as L
        from ..point_path import point_fields, _ptr, _stream
        cfg = self.point_config()
        n = self.lon_size * self.lat_size
        if x.numel() != n:
            raise ValueError('predict_grid needs all %d x %d nodes (got %d points)' % (self.lon_size, self.lat_size, x.numel()))
        heads, evec, statics = self.physics_net.field_weights(field_data, forecast_h, use_cache=use_cache)
        out_n = point_fields(cfg, input_data, heads, evec, statics, x=x, y=y, t=t)
        maps = torch.empty((6, self.lat_size, self.lon_size), dtype=torch.float32, device=out_n.device)
        ph = cfg.physics()
        L.check(L.load().dpn_grid_maps(_ptr(out_n), self.lon_size, self.lat_size, ctypes.byref(ph), int(bool(with_clip)), _ptr(maps),
                                       _stream()), 'dpn_grid_maps')
        return maps

    def training_step(self, batch: dict, optimizer, with_pde=True, max_norm=2.5e7, grad_sync=None):
        """One step body (:443-515 / :990-1065): data loss on the margin points, PDE losses on interior and margin points,
        backward, clip_grad_norm_(2.5e7), optimizer step.  `batch` holds device tensors: field_data [1,159,2405],
        forecast_h [1,1,1], margin_{x,y,t,f} [N,1], margin_data [N,6], margin_input_data [N,6], inter_{x,y,t,f} [M,1],
        inter_data [M,6].  The encoder runs once (the reference runs it three times on identical inputs)."""
        lf = self.train_cfg['losses']['loss_factor']
        self.physics_net.clear_field_cache()
        b = batch
        heads = evec = statics = meta_out = None
        if with_pde and b['field_data'].is_cuda:
            # one point pass for everything: [interior | margin] points, PDE means per group, SmoothL1 on the margin rows; the margin
            # forward serves both of its losses and all points share one backward (point_path._StepLossFn)
            from ..point_path import step_losses
            cfg = self.point_config(lf)
            meta_out = self.physics_net.encode_field(b['field_data'], b['forecast_h'], keep_embedding=grad_sync is not None)
            heads, evec, statics = self.physics_net.field_weights(b['field_data'], b['forecast_h'], meta_out=meta_out)
            cat = lambda a_, b_: torch.cat([a_.reshape(a_.shape[0], -1), b_.reshape(b_.shape[0], -1)], dim=0)
            _, inter_total, _, margin_total, data = step_losses(
                cfg, b['inter_x'].shape[0], cat(b['inter_x'], b['margin_x']), cat(b['inter_y'], b['margin_y']), cat(b['inter_t'], b['margin_t']),
                cat(b['inter_f'], b['margin_f']), cat(b['inter_data'], b['margin_input_data']), b['margin_data'], heads, evec, statics,
                beta=0.1, margin_factor=lf['margin_factor'])
            parts = {'margin_loss': data, 'inter_pde_loss': inter_total.float(), 'margin_pde_loss': margin_total.float()}
        else:
            loss = self.data_loss(b['margin_x'], b['margin_y'], b['margin_t'], b['field_data'], b['margin_input_data'], b['margin_data'],
                                  b['forecast_h'], lf['margin_factor'], use_cache=True)
            parts = {'margin_loss': loss}
            if with_pde:
                crit = nn.MSELoss()
                parts['inter_pde_loss'] = self.place_one_batch(b['inter_x'], b['inter_y'], b['inter_t'], b['inter_f'], b['field_data'],
                                                               b['inter_data'], b['forecast_h'], crit, lf, 0, 0, b['field_data'].device,
                                                               use_cache=True)
                parts['margin_pde_loss'] = self.place_one_batch(b['margin_x'], b['margin_y'], b['margin_t'], b['margin_f'], b['field_data'],
                                                                b['margin_input_data'], b['forecast_h'], crit, lf, 0, 0,
                                                                b['field_data'].device, prefix='margin', use_cache=True)
        train_loss = 0
        for v in parts.values():
            train_loss = train_loss + v
        optimizer.zero_grad()
        if getattr(self, '_seed', None) is None or self._seed.device != train_loss.device:
            self._seed = torch.ones((), dtype=train_loss.dtype, device=train_loss.device)      # persistent backward seed: no fill per step
        from ..optim import FusedClipAdam
        staged = (grad_sync is not None and hasattr(grad_sync, 'reduce_bucket') and grad_sync.active() and isinstance(optimizer, FusedClipAdam)
                  and heads is not None)
        if staged:
            # the staged form needs THIS optimiser's flat buffer laid out in the four buckets of gradient_buckets() and every parameter
            # trainable; anything else (an optimiser built without `layout`, a reducer bound to another optimiser or to none, frozen
            # parameters) takes the plain backward + grad_sync(parameters) below
            buckets = self.physics_net.gradient_buckets()
            lay = getattr(optimizer, 'layout_ids', None)
            staged = (getattr(grad_sync, 'opt', None) is optimizer and len(getattr(optimizer, 'bucket_bounds', ())) == 4
                      and lay == [[id(p) for p in b_] for b_ in buckets]
                      and all(p.requires_grad for b_ in buckets for p in b_))
        if staged:
            # data-parallel step: the backward pass is cut where a bucket of gradients is complete and that bucket's all-reduce is queued at
            # once, so it travels under the rest of the backward (what DistributedDataParallel's bucket hooks do in the reference, :903-907,
            # :1056).  Layout buckets (PhysicsNet.gradient_buckets): point statics | hyper-network heads | encoder layers | data embedding;
            # the first two travel as ONE all-reduce (the heads' backward is two launches: a collective of its own costs more than the 40 us
            # it could start earlier), the token convolution's 7.4 MB (the last gradient to complete) alone
            g = torch.autograd.grad(train_loss, [heads, evec] + list(statics), grad_outputs=self._seed)
            optimizer.place_gradients(list(statics), g[2:])
            g2 = torch.autograd.grad([heads, evec], [meta_out] + buckets[1], grad_outputs=[g[0], g[1]], allow_unused=True)
            optimizer.place_gradients(buckets[1], g2[1:])
            grad_sync.reduce_bucket(0, 2)
            x0 = getattr(self.physics_net.meta_net.model, 'last_embedding', None)
            if x0 is not None and x0.requires_grad:
                from .. import encoder_ops
                together = encoder_ops.embed_wgrad_rides_with_stack(1)     # (the stack's weight-gradient launch also wrote the token convolution's)
                g3 = torch.autograd.grad([meta_out], [x0] + buckets[2], grad_outputs=[g2[0]], allow_unused=True)
                optimizer.place_gradients(buckets[2], g3[1:])
                if not together:
                    grad_sync.reduce_bucket(2)
                g4 = torch.autograd.grad([x0], buckets[3], grad_outputs=[g3[0]], allow_unused=True)
                optimizer.place_gradients(buckets[3], g4)
                if together:
                    grad_sync.reduce_bucket(2, 4)
                else:
                    grad_sync.reduce_bucket(3)
            else:
                g3 = torch.autograd.grad([meta_out], buckets[2] + buckets[3], grad_outputs=[g2[0]], allow_unused=True)
                optimizer.place_gradients(buckets[2] + buckets[3], g3)
                grad_sync.reduce_bucket(2, 4)
            grad_sync.wait()
            self.physics_net.clear_field_cache()
        else:
            train_loss.backward(self._seed)
            self.physics_net.clear_field_cache()
            if grad_sync is not None:
                grad_sync(self.physics_net.parameters())
        if isinstance(optimizer, FusedClipAdam):                  # clip + Adam in one HIP pass
            optimizer.max_norm = float(max_norm)
            gnorm = optimizer.step()
        else:
            gnorm = torch.nn.utils.clip_grad_norm_(self.physics_net.parameters(), max_norm=max_norm)
            optimizer.step()
        return train_loss.detach(), {k: v.detach() for k, v in parts.items()}, gnorm


    # ------------------------------------------------------------------ training loops (:334-846, :848-1404; step body only)
    def build_optimizer(self, **overrides):
        """Fused clip + Adam over the PhysicsNet with the config's optimiser settings (cfg:151-155; `initial_lr` as :394 sets it), its flat
        gradient buffer laid out in backward-completion order (PhysicsNet.gradient_buckets)."""
        from ..optim import FusedClipAdam
        oc = dict(self.train_cfg.get('optimizer', {}))
        name = oc.pop('name', 'Adam')
        if name != 'Adam':
            raise NotImplementedError('the fused optimiser implements Adam (cfg:151-155); got %r' % name)
        oc.update(overrides)
        opt = FusedClipAdam(self.physics_net.parameters(), layout=self.physics_net.gradient_buckets(), **oc)
        opt.param_groups[0].setdefault('initial_lr', opt.param_groups[0]['lr'])
        return opt

    def _build_lr_schedule(self, optimizer, current_epoch):
        sc = dict(self.train_cfg.get('lr_schedule') or {})
        if not sc:
            return None
        name = sc.pop('name', 'CosineAnnealingLR')
        sc.pop('verbose', None)                       # cfg:160-165 passes verbose=True, which torch >= 2.7 rejects (defect 4)
        return getattr(torch.optim.lr_scheduler, name)(optimizer, last_epoch=current_epoch - 1, **sc)

    def _train_samples(self, kwargs, epoch):
        src = kwargs.get('samples', self.train_cfg.get('train_data', {}).get('samples'))
        if src is None:
            # the reference's PhysicsDataset reads GeoTIFF / xarray files (dataset/physics_dataset.py: file I/O outside this build, SURVEY.md
            # section 2 row 10) and fails when they are missing; so does this loop.  Synthetic data is an explicit choice.
            raise RuntimeError("run_train_interface: no `samples` source (a sequence / callable of training batches, as keyword or as "
                               "train_cfg['train_data']['samples']).  samples='synthetic' (train.py --synthetic) draws random field samples "
                               "and on-device collocation batches instead -- noise, for smoke runs only")
        if isinstance(src, str):
            if src != 'synthetic':
                raise ValueError("samples=%r: the only named source is 'synthetic'" % src)
            if getattr(self, '_synthetic_samples', None) is None:
                from ..sampler import SyntheticSamples
                td = self.train_cfg.get('train_data', {})
                dev = next(self.physics_net.parameters()).device
                self._synthetic_samples = SyntheticSamples(dev, n_margin=td.get('label_batch_size', 20480), n_inter=td.get('batch_size_inter', 4096),
                                                           leads=int(kwargs.get('samples_per_epoch', 61)), lat=self.lat_size, lon=self.lon_size)
                print("run_train_interface: samples='synthetic' -- random field samples (%d per epoch) and on-device collocation batches; "
                      "the checkpoints of this run are trained on noise" % len(self._synthetic_samples))
            return self._synthetic_samples
        return src(epoch) if callable(src) else src

    @staticmethod
    def _shard_samples(samples, rank, world, shuffle_seed=None):
        """DistributedSampler semantics (:936, drop_last=False): every rank takes ceil(n / world) samples of the epoch -- rank r the samples
        r, r + world, ... -- and the tail wraps around to the epoch's first samples, so that all ranks run the same number of steps (and
        of all-reduces).  A sequence is indexed (a rank touches only its own samples); any other iterable is consumed round by round.
        shuffle_seed (sequences only): the order is `torch.randperm(n)` drawn from a generator seeded with it, as DistributedSampler(shuffle=True,
        seed=0) does; the reference never calls `set_epoch`, so every epoch walks the SAME seed-0 permutation (torch/utils/data/distributed.py)."""
        if hasattr(samples, '__len__') and hasattr(samples, '__getitem__'):
            n = len(samples)
            if n == 0:
                return
            if shuffle_seed is not None:
                g = torch.Generator()
                g.manual_seed(int(shuffle_seed))
                order = torch.randperm(n, generator=g).tolist()
            else:
                order = list(range(n))
            if world == 1:
                for i in order:
                    yield samples[i]
                return
            for k in range(-(-n // world)):
                yield samples[order[(rank + k * world) % n]]
            return
        if world == 1:
            yield from samples
            return
        head, buf = [], []
        for smp in samples:
            if len(head) < world:
                head.append(smp)
            buf.append(smp)
            if len(buf) == world:
                yield buf[rank]
                buf = []
        if buf:                                          # incomplete last round: padded with the epoch's first samples
            k = 0
            while len(buf) < world:
                buf.append(head[k % len(head)])
                k += 1
            yield buf[rank]

    def _epoch_samples(self, kwargs, epoch, rank, world, dist_mode=False):
        """The epoch's samples of this rank.  `samples` may be a callable samples(epoch) -> all samples, sharded here like DistributedSampler.
        The distributed loop shuffles an indexable source the way the reference's sampler does (:936: DistributedSampler's defaults = a seed-0
        permutation, the same every epoch because `set_epoch` is never called); keyword `shuffle=False` keeps the given order, `shuffle_seed`
        changes the seed.  The single-process loop takes the samples in the given order (the reference's DataLoader(shuffle=True), :419, draws an
        unseeded permutation per epoch: shuffle in the source if wanted).  Iterables that cannot be indexed are consumed in their own order.
        With samples_per_rank=True (keyword, or attribute `samples.per_rank = True`) the callable is samples(epoch, rank, world) -> this rank's
        samples only (nothing is drawn for the other ranks)."""
        src = kwargs.get('samples', self.train_cfg.get('train_data', {}).get('samples'))
        per_rank = kwargs.get('samples_per_rank', getattr(src, 'per_rank', False))
        if callable(src) and per_rank:                   # explicit protocol (keyword samples_per_rank=True or attribute samples.per_rank)
            return src(epoch, rank, world)
        seed = int(kwargs.get('shuffle_seed', 0)) if (dist_mode and kwargs.get('shuffle', True)) else None
        return self._shard_samples(self._train_samples(kwargs, epoch), rank, world, shuffle_seed=seed)

    def _run_train(self, dist_mode, **kwargs):
        tc = self.train_cfg
        num_epoch = int(kwargs.get('num_epoch', tc['num_epoch']))
        self.dx, self.dy = float(tc['dx']), float(tc['dy'])
        time_step = tc.get('lable_time_step', 1)
        self.dt = float(60 * 60 * time_step)
        checkpoint_path = kwargs.get('checkpoint_path') or tc.get('checkpoints', {}).get('checkpoints_path')
        save_step = int(tc.get('checkpoints', {}).get('save_step', 1))
        log_step = int(tc.get('log', {}).get('log_step', 100))
        pde_start = int(kwargs.get('pde_start_step', 2000))              # :436-441: PDE losses switch on at global_step >= 2000
        max_steps = kwargs.get('max_steps')
        with_pde_cfg = bool(tc.get('with_pde', True)) if dist_mode else True      # only the _dist loop reads train_cfg['with_pde'] (:857)
        rank, world, sync = 0, 1, None
        if dist_mode:
            from .. import distributed as D
            rank, world, local = D.init_from_env(kwargs.get('backend'))
            device = torch.device('cuda', local if kwargs.get('device') is None else kwargs['device'])
            torch.cuda.set_device(device)
        else:
            device = torch.device(kwargs.get('device', tc.get('device', 'cuda:0')))
        self.physics_net.to(device)
        self.pe.to(device)
        if checkpoint_path and rank == 0:
            os.makedirs(checkpoint_path, exist_ok=True)
        loss_factor = tc['losses']['loss_factor']
        state_dict, current_epoch, global_step = (None, 0, 0)
        if checkpoint_path:
            state_dict, current_epoch, global_step = self.load_model(checkpoint_path, prefix='physics', map_location=device)
        if state_dict is not None:
            if rank == 0:
                print('resume from epoch %d global_step %d' % (current_epoch, global_step))
            self.physics_net.load_state_dict(state_dict['model'], strict=True)
        optimizer = self.build_optimizer()
        if dist_mode:
            D.broadcast_parameters(self.physics_net)                    # DistributedDataParallel does this at wrap time (:903-907)
            sync = D.GradientAllReduce(optimizer)
        lr_schedule = self._build_lr_schedule(optimizer, current_epoch)
        self.physics_net.train()
        last = None
        for epoch in range(current_epoch, num_epoch):
            for batch in self._epoch_samples(kwargs, epoch, rank, world, dist_mode):   # DistributedSampler (:936): one field sample per rank per step
                with_pde = with_pde_cfg and global_step >= pde_start
                self.with_clip = True
                global_step += 1
                batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
                loss, parts, gnorm = self.training_step(batch, optimizer, with_pde=with_pde, grad_sync=sync)
                last = {'loss': loss, 'parts': parts, 'grad_norm': gnorm}
                if rank == 0 and global_step % log_step == 1:
                    print('epoch %d step %d loss %.6g %s' % (epoch, global_step, float(loss),
                                                              ' '.join('%s %.4g' % (k, float(v)) for k, v in parts.items())))
                if global_step % log_step == 1:                       # (the loop synchronises here anyway: float(loss))
                    from ..encoder_ops import check_enc_status
                    check_enc_status()                                # an encoder weight outside the f16 hi+lo split's range raises HERE, named
                if max_steps is not None and global_step >= max_steps:
                    break
            if epoch % save_step == 0:
                if lr_schedule is not None:
                    lr_schedule.step()
                    optimizer.sync_hyper()
                from ..encoder_ops import check_enc_status
                check_enc_status()
                if checkpoint_path and rank == 0:
                    self.save_model(checkpoint_path, epoch, global_step, prefix='physics', dx=self.dx, dy=self.dy, dt=self.dt,
                                    pred_x_span=self.dx * self.lon_size, pred_y_span=self.dy * self.lat_size, pred_t_span=self.pred_t_span,
                                    label_time_step=time_step, obs_norm_cfg=self.obs_norm_cfg)
            if max_steps is not None and global_step >= max_steps:
                break
        return {'epoch': epoch if num_epoch > current_epoch else current_epoch, 'global_step': global_step, 'optimizer': optimizer,
                'lr': optimizer.param_groups[0]['lr'], 'last': last}

    def run_train_interface(self, **kwargs):
        """The single-GPU training loop (:334-846) reduced to what is on the path: per-step body (:443-515) = training_step, PDE losses on
        from global_step >= 2000, CosineAnnealingLR stepped and a checkpoint written once per epoch (:831-845), resume from
        `physics_latest.pth` (:389-397).  kwargs of the reference: checkpoint_path, log_path (unused: no tensorboard / JPEG output here);
        added: samples (see _train_samples), num_epoch, max_steps, pde_start_step, device."""
        return self._run_train(False, **kwargs)

    def run_train_interface_dist(self, **kwargs):
        """The data-parallel loop (:848-1404): one process per GPU (start it with torchrun; torch.distributed is initialised from the
        environment, which the reference never does -- SURVEY section 0, defect 2), rank r takes every world-th sample
        (DistributedSampler, :936), gradients are averaged by distributed.GradientAllReduce on the optimiser's flat gradient buffer
        (replaces the DistributedDataParallel wrap :903-907), rank 0 writes the checkpoints."""
        return self._run_train(True, **kwargs)


class StagedPdeStep:
    """place_one_batch (or, lead_batch, place_lead_batch) + backward, cut where a bucket of gradients is complete
    (PhysicsNet.gradient_buckets), so that a data-parallel caller can start that bucket's all-reduce while the rest of the backward pass runs
    (BASELINE configs[3]: "bucketed overlap with backward"; the reference gets the same from DistributedDataParallel's bucket hooks,
    interface_physics.py:903-907,:1056):
        stages[0]  zero_grad, encoder + heads forward, point forward / residuals, point backward,
                   hyper-network heads backward                     -> layout buckets 0 and 1 (48 static tensors, the heads): stage_buckets[0] = (0, 2)
        stages[1]  encoder layers + norm + projection backward       -> layout bucket 2: stage_buckets[1] = (2, 3)
        stages[2]  data embedding backward (the token convolution's 7.4 MB gradient: the last to complete, it travels alone so that the
                   other 6.4 MB of the encoder start one launch earlier) -> layout bucket 3: stage_buckets[2] = (3, 4)
    One field on the fused encoder (round 6): stages[1] and stages[2] are ONE stage and buckets 2, 3 one all-reduce -- the token convolution's gradient is
    written by the encoder stack's own weight-gradient launch, so both buckets complete together (encoder_ops.embed_wgrad_rides_with_stack).
    Each stage is a plain callable (capturable in a hipGraph of its own, on one capture stream and one memory pool); after stage i the caller
    queues `grad_sync.reduce_bucket(*stage_buckets[i])`.  (Round 2 cut the heads' backward off as a stage of its own: three collectives and
    four graph segments cost 1.80 -> 2.00 ms with a one-rank RCCL group; the heads' backward is two launches, so the statics' all-reduce
    loses ~40 us of head start and still has the whole encoder backward, 0.3 ms, to travel under.)
    batch: place_one_batch's tensors (x, y, t, f, field_data, coord_data, forecast_h); with lead_batch they carry a leading B (field_data
    [B, 159, 2405], x .. f [B, N], coord_data [B, N, 6], forecast_h [B, 1, 1]) and the loss is the mean of the B field totals."""

    def __init__(self, interface, optimizer, batch, loss_factor=None, lead_batch=False):
        self.m, self.opt, self.b = interface, optimizer, batch
        self.lf = loss_factor or interface.train_cfg['losses']['loss_factor']
        self.lead_batch = bool(lead_batch)
        net = interface.physics_net
        self.buckets = net.gradient_buckets()
        self.loss = None
        self.stages = (self.stage_points_and_heads, self.stage_encoder, self.stage_embedding)
        self.stage_buckets = ((0, 2), (2, 3), (3, 4))
        from .. import encoder_ops
        if not self.lead_batch and encoder_ops.embed_wgrad_rides_with_stack(1):
            # Round 6: one field on the fused encoder -- the token convolution's weight gradient is a problem of the stack's ONE weight-gradient launch,
            # the embedding's own backward launches nothing: buckets 2 and 3 complete together.  Two all-reduces behind each other (and a graph
            # segment that is empty) were 73 us of exposed tail on a one-rank RCCL group (gap 21 + 10 + gap 17 + 8 + gap 17); one is gap + 16 + gap.
            self.stages = (self.stage_points_and_heads, self.stage_encoder_and_embedding)
            self.stage_buckets = ((0, 2), (2, 4))
        self._seed = None

    def _assign(self, params, grads):
        self.opt.place_gradients(params, grads)          # the rare gradient that did not land in its slot is copied there before the all-reduce

    def stage_points(self):
        m, b = self.m, self.b
        net = m.physics_net
        self.opt.zero_grad(set_to_none=True)
        cfg = m.point_config(self.lf)
        self.meta_out = net.encode_field(b['field_data'], b['forecast_h'], keep_embedding=True)
        self.x0 = getattr(net.meta_net.model, 'last_embedding', None)        # the cut between stages 1 and 2 (None: the encoder ran unfused)
        object.__setattr__(net.meta_net.model, 'last_embedding', None)       # (this object holds it from here on)
        self.heads, self.evec, statics = net.field_weights(b['field_data'], b['forecast_h'], meta_out=self.meta_out)
        if self.lead_batch:
            from ..point_path import pde_losses_batch
            B = b['field_data'].shape[0]
            self.heads, self.evec = self.heads.reshape(B, 256, -1), self.evec.reshape(B, 6, 256)
            _, totals = pde_losses_batch(cfg, b['x'], b['y'], b['t'], b['f'], b['coord_data'], self.heads, self.evec, statics)
            total = totals.mean().float()
        else:
            _, total = pde_losses(cfg, b['x'], b['y'], b['t'], b['f'], b['coord_data'], self.heads, self.evec, statics, with_total=True)
        if self._seed is None:
            self._seed = torch.ones((), dtype=total.dtype, device=total.device)
        g = torch.autograd.grad(total, [self.heads, self.evec] + list(statics), grad_outputs=self._seed)
        self.g_heads, self.g_evec = g[0], g[1]
        self._assign(statics, g[2:])
        self.loss = total.detach()
        return self.loss

    def stage_points_and_heads(self):
        loss = self.stage_points()
        self.stage_heads()
        return loss

    def stage_heads(self):
        params = self.buckets[1]
        g = torch.autograd.grad([self.heads, self.evec], [self.meta_out] + params, grad_outputs=[self.g_heads, self.g_evec], allow_unused=True)
        self.g_meta = g[0]
        self._assign(params, g[1:])

    def stage_encoder(self):
        if self.x0 is None or not self.x0.requires_grad:             # no cut available: the whole encoder in this stage, stage 2 is empty
            params = self.buckets[2] + self.buckets[3]
            g = torch.autograd.grad([self.meta_out], params, grad_outputs=[self.g_meta], allow_unused=True)
            self._assign(params, g)
            self.g_x0 = None
            return
        params = self.buckets[2]
        g = torch.autograd.grad([self.meta_out], [self.x0] + params, grad_outputs=[self.g_meta], allow_unused=True)
        self.g_x0 = g[0]
        self._assign(params, g[1:])

    def stage_encoder_and_embedding(self):
        self.stage_encoder()
        self.stage_embedding()

    def stage_embedding(self):
        if self.g_x0 is not None:
            params = self.buckets[3]
            g = torch.autograd.grad([self.x0], params, grad_outputs=[self.g_x0], allow_unused=True)
            self._assign(params, g)
        self.meta_out = self.heads = self.evec = self.g_heads = self.g_evec = self.g_meta = self.x0 = self.g_x0 = None
        self.m.physics_net.clear_field_cache()

"""Constants of the one configuration the reference ships (configs/DeepPhysiNet_NCEP_cfg.py), restated as the
inputs of this build: model sizes (:11-32), observation normalisation + clip bounds (:64-76), grid (:10,:93-95),
loss factors (:137-148), optimiser (:151-155)."""
import copy

IMG_SIZE = (145, 257)          # (lat, lon) of the 0.25 degree grid

_NCEP = dict(
    name='InterfacePhysics',
    meta_cfg=dict(name='TransformerNet', enc_in=2405, c_out=256, d_model=256, n_heads=8, e_layers=4, d_ff=256, dropout=0.5,
                  activation='gelu', output_attention=False),
    net_cfg=dict(name='PhysicsNet', in_channels=192, hidden_channels=256, out_channels=1, token_num=155 + 4, learnable_token_num=256),
    variable_cfg=dict(),
    obs_norm_cfg=dict(
        pres=dict(name='PSFC', norm_factor=[89741.36105771353, 13296.749084125422], norm_type='mean_norm', bound=[10000, 500000], use_norm=True),
        t2=dict(name='t2', norm_factor=[283.58054561520305, 15.583177935722373], norm_type='mean_norm', bound=[50, 500], use_norm=True),
        u10=dict(name='u10', norm_factor=[0.14507186950562942, 3.0050219075895894], norm_type='mean_norm', bound=[-500, 500], use_norm=True),
        v10=dict(name='v10', norm_factor=[-0.17325370241478535, 3.006602165591562], norm_type='mean_norm', bound=[-500, 500], use_norm=True),
        q2=dict(name='q2', norm_factor=[0.007909478276582905, 0.006304067969976075], norm_type='mean_norm', bound=[1e-6, 10], use_norm=True),
        rio=dict(name='rio', norm_factor=[1.0966503643401704, 0.15166081218127583], norm_type='mean_norm', bound=[1e-6, 10], use_norm=True),
    ),
    train_cfg=dict(
        batch_size=1, device='cuda:0', num_epoch=201, with_pde=True, lable_time_step=1, dx=27000, dy=27000, img_size=IMG_SIZE,
        train_data=dict(input_time_step=6, input_time_step_nums=4, forecast_time_period=360, label_time_step=1,
                        label_img_size=IMG_SIZE, label_batch_size=2048 * 10, batch_size_inter=2048 * 2),
        losses=dict(pde_loss=dict(name='MSELoss'), prediction_loss=dict(name='WeightSmoothL1Loss', beta=0.1),
                    loss_factor=dict(sample_factor=1.e6, margin_factor=1.e6, motion_u_factor=1.e3, motion_v_factor=1.e3,
                                     continuous_factor=1.e10, energy_factor=1e1, vapor_factor=1.e14, gas_factor=1.e-7)),
        optimizer=dict(name='Adam', lr=1e-4, weight_decay=1e-4),
        lr_schedule=dict(name='CosineAnnealingLR', T_max=5, eta_min=5e-6),
    ),
    test_cfg=dict(),
    inference_cfg=dict(),
)


def ncep_config(img_size=IMG_SIZE, dx=27000, dy=27000):
    """A fresh copy of the NCEP configuration; img_size=(37,65), dx=dy=108000 gives the 1-degree plumbing case."""
    cfg = copy.deepcopy(_NCEP)
    cfg['train_cfg']['img_size'] = tuple(img_size)
    cfg['train_cfg']['dx'], cfg['train_cfg']['dy'] = dx, dy
    return cfg

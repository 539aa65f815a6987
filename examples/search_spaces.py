"""Search spaces of the two reference workflows, as plain dicts of mitune domains.

``SAMPLE_SPACE``  <-> /root/reference/ray-tune-hpo-regression-sample.py:140-147 (6 keys)
``FULL_SPACE``    <-> /root/reference/ray-tune-hpo-regression.py:379-400 (20 keys, incl. the dependent
                      ``dim_feedforward = d_model * choice([2, 3, 4])``, which works here because a
                      ``sample_from`` lambda may return a domain)
``MLP_SPACE``     -   the packed-MLP search used by the 256-trial / 1024-trial configurations
"""
from mitune import tune

SAMPLE_SPACE = {
    "d_model": tune.choice([32, 64, 128]),
    "num_heads": tune.choice([2, 4]),
    "num_layers": tune.choice([1, 2, 3]),
    "dropout": tune.uniform(0.1, 0.5),
    "learning_rate": tune.loguniform(1e-4, 1e-2),
    "weight_decay": tune.loguniform(1e-6, 1e-2),
}

FULL_SPACE = {
    "num_heads": tune.choice([2, 4, 8, 16, 32]),
    "num_encoder_layers": tune.choice([2, 4, 6, 8, 12, 16]),
    "d_model": tune.choice([64, 128, 192, 256, 320, 512]),
    "dim_feedforward": tune.sample_from(lambda spec: spec.config.d_model * tune.choice([2, 3, 4])),
    "dropout": tune.loguniform(0.01, 0.5),
    "learning_rate": tune.loguniform(1e-5, 5e-2),
    "batch_size": tune.choice([16, 32, 64, 128, 256]),
    "weight_decay": tune.loguniform(1e-6, 1e-1),
    "max_seq_length": tune.choice([50, 100, 200, 500, 1000, 2000]),
    "attention_type": tune.choice(["scaled_dot_product", "linear_attention", "multi_head_attention"]),
    "key_dim_scaling": tune.choice([1.0, 0.5, 0.25]),
    "attn_kernel_size": tune.choice([3, 5, 7]),
    "depthwise_separable_conv": tune.choice([True, False]),
    "shared_weights": tune.choice([True, False]),
    "stochastic_depth_rate": tune.uniform(0.0, 0.2),
    "warmup_steps": tune.choice([100, 500, 1000, 2000]),
    "total_steps": tune.choice([10000, 20000, 50000, 100000]),
    "gradient_clipping": tune.uniform(0.0, 1.0),
    "loss_function": tune.choice(["mse", "mae", "huber", "mape"]),
    "optimizer": tune.choice(["adam", "adamw", "sgd", "rmsprop"]),
}

MLP_SPACE = {
    "width_0": tune.choice([64, 128, 256, 512]),
    "width_1": tune.choice([64, 128, 256, 512]),
    "width_2": tune.choice([64, 128, 256, 512]),
    "num_layers": 3,
    "batch_size": tune.choice([256, 512, 1024]),
    "learning_rate": tune.loguniform(1e-4, 1e-2),
    "weight_decay": tune.loguniform(1e-6, 1e-2),
    "optimizer": "adamw",
    "loss_function": "mse",
}

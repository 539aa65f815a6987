import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_xf_engine_gpu as T
from mitune.models.transformer import TransformerModel
from mitune.packed.permutation import permute, round_keys
from mitune.train.losses import get_criterion
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tr = T._trainable(); tr.model_kwargs["max_batch"] = 64
cfg = dict(T.CONFIGS[which], optimizer="sgd", momentum=0.0, learning_rate=1.0, weight_decay=0.0, gradient_clipping=0.0,
           warmup_steps=0, total_steps=0, batch_size=64)
eng = T._xf(tr); eng.load(1, 10 + which, cfg)
before = eng.export(1)["model"]; eng.train_epoch(); torch.cuda.synchronize(); after = eng.export(1)["model"]
X, y = tr.train
model = TransformerModel.from_config(cfg, T.F).double(); model.load_state_dict({k: v.double() for k, v in before.items()}); model.train()
idx = torch.from_numpy(permute(np.arange(64, dtype=np.uint32), 64, round_keys(3, 10 + which, 0)).astype(np.int64))
loss = get_criterion(cfg["loss_function"])(model(X[idx].double()).reshape(-1), y[idx].double()); loss.backward()
for name, p in model.named_parameters():
    got = (before[name] - after[name]).double(); ref = p.grad
    print(f"{name:50s} err {(got-ref).abs().max().item():.3e}  ref {ref.abs().max().item():.3e}  got {got.abs().max().item():.3e}")
name = "layers.0.feedforward.0.weight" if "layers.0.feedforward.0.weight" in before else "layers.0.feedforward.pointwise.weight"
got = (before[name] - after[name]).double().reshape(model.state_dict()[name].shape[0], -1); ref = dict(model.named_parameters())[name].grad.reshape(got.shape)
bad = ((got - ref).abs() > 1e-6)
print("shape", tuple(got.shape), "bad elements", int(bad.sum()), "bad rows", bad.any(1).nonzero().flatten().tolist()[:40], "bad cols", bad.any(0).nonzero().flatten().tolist()[:70])
r = bad.any(1).nonzero().flatten()[0].item()
print("row", r, "got", got[r, :8].tolist(), "ref", ref[r, :8].tolist())
lay = eng.slots[1].layout
print({k: v for k, v in lay.off.items() if k.startswith("layers.0") or k.startswith("input")})

"""Drop-in evidence for EVERY recipe of the reference (SURVEY 8b "run.sh recipes are drop-in"): for each egs/*/*/conf/*.yaml that names
a model of the hot path, instantiate the REFERENCE's model class with the file's model_params in this container and record

    recipe path, model_type, model_params, trainer_type, collater_type, criterions (+ their params), optimizer / scheduler settings,
    the reference model's state_dict as [key, shape, dtype] triples and its trainable parameter count

into tests/golden/recipe_configs.json (data: configuration values and tensor shapes, no source).  tests/test_oracle_golden.py builds
the product's model from the same params on the CPU and requires identical keys, shapes and dtypes, and that the trainer / collater /
criterion names resolve in seq2seq_vc_amd.  Runs only where /root/reference exists.

    python tools/gen_recipe_configs.py
"""
import glob
import json
import os
import sys

import yaml

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gen_golden import REF, import_reference  # noqa: E402

import torch  # noqa: E402


def main():
    M, L, _ = import_reference()
    out = []
    for path in sorted(glob.glob(os.path.join(REF, "egs", "*", "*", "conf", "*.yaml"))):
        with open(path) as f:
            cfg = yaml.safe_load(f)
        mt = cfg.get("model_type")
        if mt is None or not hasattr(M, mt):
            continue
        params = dict(cfg["model_params"])
        extra = {}
        if mt == "TransformerTTS" and "idim" not in params:      # injected from the token list at bin/tts_train.py
            extra = {"idim": 78}
        torch.manual_seed(0)
        model = getattr(M, mt)(**params, **extra)
        sd = model.state_dict()
        rec = {"recipe": os.path.relpath(path, REF), "model_type": mt, "model_params": params, "injected_params": extra,
               "trainer_type": cfg.get("trainer_type"), "collater_type": cfg.get("collater_type"),
               "criterions": cfg.get("criterions"), "optimizer_type": cfg.get("optimizer_type"),
               "optimizer_params": cfg.get("optimizer_params"), "scheduler": cfg.get("scheduler"),
               "scheduler_params": cfg.get("scheduler_params"), "grad_norm": cfg.get("grad_norm"),
               "gradient_accumulate_steps": cfg.get("gradient_accumulate_steps"), "batch_size": cfg.get("batch_size"),
               "lambda_align": cfg.get("lambda_align"), "init_mods": cfg.get("init-mods"), "freeze_mods": cfg.get("freeze-mods"),
               "n_trainable": sum(p.numel() for p in model.parameters() if p.requires_grad),
               "state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]}
        out.append(rec)
        print(f"{rec['recipe']}: {mt}, {len(sd)} keys, {rec['n_trainable'] / 1e6:.2f} M parameters, trainer {rec['trainer_type']}, "
              f"collater {rec['collater_type']}, criterions {list((rec['criterions'] or {}).keys())}")
    dst = os.path.join(ROOT, "tests", "golden", "recipe_configs.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"wrote {dst}: {len(out)} recipes, {os.path.getsize(dst) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()

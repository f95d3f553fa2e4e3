"""Golden vectors for the data-side transform (RLDSBatchTransform vla/datasets/datasets.py:30-185, ActionTokenizer
vla/action_tokenizer.py, PurePromptBuilder base_prompter.py:27-79) from the REAL reference classes -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_transform.py

The Llama tokenizer and the CLIP processor are not in the image: oracle/recipe.py's ToyTokenizer / ToyImageTransform stand in for them
on BOTH sides (what is pinned is the transform's own logic: prompt text, token order, label masking, tactile / mask-channel handling).
Writes tests/golden/transform.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402


def main():
    ref_import.setup()
    from models.backbones.llm.prompting import PurePromptBuilder
    from vla.action_tokenizer import ActionTokenizer
    from vla.datasets.datasets import RLDSBatchTransform
    tok = recipe.ToyTokenizer()
    res = {}
    for name, kw in recipe.RLDS_CASES.items():
        at = None if name == "no_action_tok" else ActionTokenizer(tok)
        tf = RLDSBatchTransform(at, tok, recipe.ToyImageTransform(), PurePromptBuilder, predict_stop_token=name != "no_stop",
                                use_pointcloud=kw.get("with_pc", False), use_tactile=kw.get("with_tactile", False))
        out = tf(recipe.make_rlds_sample(**kw))
        for k, v in out.items():
            if k == "images":
                for cam, t in v.items():
                    res[f"{name}::images.{cam}"] = t[:, ::37, ::41].numpy()
            elif torch.is_tensor(v):
                res[f"{name}::{k}"] = (v[:, ::37, ::41] if k == "next_images" else v).numpy()
            elif v is None:
                res[f"{name}::{k}"] = np.array("None")
    at = ActionTokenizer(tok)
    a = np.linspace(-1.5, 1.5, 1001)
    res["at_ids"] = tok.vocab_size - np.digitize(np.clip(a, -1.0, 1.0), at.bins)
    res["at_str"] = np.array(at(a[::100]))
    res["at_decode"] = at.decode_token_ids_to_actions(np.arange(tok.vocab_size - 257, tok.vocab_size))
    res["at_begin"] = np.array(at.action_token_begin_idx)
    pb = PurePromptBuilder("openvla")
    pb.add_turn("human", " <image> What now? ")
    pb.add_turn("gpt", "")
    res["pb_prompt"] = np.array(pb.get_prompt())
    res["pb_potential"] = np.array(pb.get_potential_prompt("and then?"))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "transform.npz"), **res)
    print("transform.npz:", len(res), "arrays; plain ids", res["plain::input_ids"][-12:], "labels", res["plain::labels"][-12:])


if __name__ == "__main__":
    main()

"""GPU probe: end-to-end generate on the flat (x5) lm_head case vs the reference tokens (informational)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_util as G
from tests.test_oracle_golden import GEN_RUNS, gen_key, generate_case_weights
from oracle.make_golden_cases import GENERATE_CASE as C
from slam_llm_amd.model import SlamHipModel

dev = torch.device("cuda:0")
fx = G.load("generate")
for scale in (24.0, 5.0):
    W = generate_case_weights(scale)
    model = SlamHipModel(dict(C["cfg"], lora_dropout=0.0), dev).load_weights(W)
    model.eval()
    b = {k[len("batch."):]: torch.from_numpy(fx[k]).to(dev) for k in fx.files if k.startswith("batch.")}
    eos = int(fx[f"s{scale}.eos"])
    for nb, lp, pad, rp in GEN_RUNS:
        got = model.generate(**{k: v.clone() for k, v in b.items()}, max_new_tokens=C["max_new_tokens"], num_beams=nb,
                             length_penalty=lp, eos_token_id=eos, pad_token_id=pad, repetition_penalty=rp).cpu().numpy()
        want = fx[gen_key(scale, nb, lp, pad, rp)]
        ok = got.shape == want.shape and (got == want).all()
        print(scale, nb, lp, pad, "MATCH" if ok else f"DIFF\n{got}\n{want}")

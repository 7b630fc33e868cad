import sys, os, copy, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_model_gpu as T
from hqq_amd.backends.hip import group_llama_projections
from hqq_amd.core.quantize import BaseQuantizeConfig
from hqq_amd.utils.generation import GraphedGreedyDecoder
from hqq_amd.utils.model import quantize_model
from hqq_amd.utils.patching import prepare_for_inference
nbits = int(sys.argv[1]); out = sys.argv[2]
model = T._tiny_llama()
quantize_model(model, BaseQuantizeConfig(nbits=nbits, group_size=64, axis=1), compute_dtype=torch.float16, device="cuda")
ids = torch.randint(0, 512, (1, 6), generator=torch.Generator().manual_seed(3)).cuda()
prepare_for_inference(model, backend="hip")
group_llama_projections(model)
dec = GraphedGreedyDecoder(model, max_cache_len=64, glue="folded")
got = dec.generate(ids, 32, use_graph=False)
# per-step logits, teacher-forced on `got`
dec2 = GraphedGreedyDecoder(model, max_cache_len=64, glue="folded")
logs = []
orig = dec2._pick
torch.save({"tokens": got.cpu()}, out)
print(got.tolist())

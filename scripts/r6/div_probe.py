import json, os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("UMBRELLA_SYNTHETIC", "1")
import __graft_entry__ as ge; ge.build()
from helpers import load_golden
from hip_helpers import static_engine
g = load_golden(); dev = torch.device("cuda:0")
case_name = "static_3x4_selfdraft_stochastic"
case = json.load(open("/root/repo/tests/golden/engines_stochastic.json"))["cases"][case_name]
c = case["config"]
eng, _ = static_engine(g, dev, torch.float16, self_draft=True, hip_graph=False, max_length=c["max_length"], safe_buffer=c["safe_buffer"], eos=tuple(case["eos"]),
                       temperature=c["temperature"], topp=c["topp"], topk=c["topk"], repetition_penalty=c["repetition_penalty"], uniform_samples=torch.tensor(case["uniform_samples"]))
assert eng._prefill(torch.tensor([case["prompt"]]))
for i, rec in enumerate(case["iters"]):
    if eng.num_nodes != rec["n"]: print("n differs", i); break
    eng.build_tree(); tree = eng.tokens[rec["n"]:rec["n"] + eng.tree_size].tolist()
    eng._verify_forward(); eng._sample(); sampled = eng.sampled.tolist(); eng._commit(); go = eng._finish_iteration()
    bad = [k for k, ok in (("tree", tree == rec["tree_tokens"]), ("sampled", sampled == rec["sampled"]), ("num_nodes", eng.num_nodes == rec["num_nodes"]), ("bonus", int(eng.tokens[eng.num_nodes]) == rec["bonus"]), ("go", go == rec["go_on"])) if not ok]
    if bad:
        print("iter", i, "differs in", bad)
        print("tree diff idx", [j for j in range(len(tree)) if tree[j] != rec["tree_tokens"][j]])
        print("sampled diff idx", [j for j in range(len(sampled)) if sampled[j] != rec["sampled"][j]])
        print(rec.keys())
        break

"""The stream assignment ROCm's graph executor gives the captured training step: run with DEBUG_HIP_GRAPH_DOT_PRINT=1 from a
scratch directory (the runtime writes graph_*_dot_print_* there), then this script parses the newest file: one line per kernel
node -- internal stream, kernel, predecessors.  `python tools/graph_dot.py capture` captures one step (writes the file);
`python tools/graph_dot.py parse <file>` prints the table."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def capture():
    import torch
    import bench
    from frustum_convnet_amd.train_state import FlatTrainState
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    model.defer_metrics_join = os.environ.get("FCN_IOU_JOIN", "late") != "early"
    state = FlatTrainState(model, lr=1e-3, weight_decay=1e-4)
    data = bench.make_data("car", 32, 1024, 1234, dev)
    prefetch = os.environ.get("FCN_PREFETCH", "1") != "0"

    def step():
        losses, _ = model(data)
        if prefetch:
            model.prefetch(data)
        model.backward(losses["total_loss"])
        state.adam_step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay()
    torch.cuda.synchronize()


def parse(path, only=None):
    txt = open(path).read()
    nodes = re.findall(r'"graph_\d+_node_(\d+)"\[[^\]]*?label="(\d+)\n([^\n]+)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', txt)
    dem = subprocess.run(["c++filt"], input="\n".join(n[2] for n in nodes), capture_output=True, text=True).stdout.strip().split("\n")
    pred = {}
    for a, b in re.findall(r'"graph_\d+_node_(\d+)"\s*->\s*"graph_\d+_node_(\d+)"', txt):
        pred.setdefault(int(b), []).append(int(a))
    for (nid, lab, nm, sid, sig), d in zip(nodes, dem):
        d = re.sub(r"\(.*", "", d).replace("void ", "")[:44]
        if only and not any(k in d for k in only):
            continue
        print("%3s s%s %-4s %-46s <- %s" % (nid, sid, "SIG" if sig == "true" else "", d, pred.get(int(nid), [])))


if __name__ == "__main__":
    if sys.argv[1] == "capture":
        capture()
    else:
        parse(sys.argv[2], sys.argv[3:] or None)

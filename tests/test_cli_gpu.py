"""The command-line driver end to end on the MI355X: a cora-shaped synthetic graph, both step engines."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_runs_reference_style_experiment(capsys):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import main as cli
    f1 = cli.main(["--dataset", "cora", "--max_epochs", "3", "--runs", "1", "--eval_frequency", "2", "--batch_size", "64",
                   "--num_samples", "16", "--seed", "1", "--e_cap", "16384", "--hidden_dim", "64"])
    out = capsys.readouterr().out
    assert 0.0 <= f1 <= 1.0 and "valid_accuracy=" in out and "test_accuracy=" in out and "Acc:" in out
    # random sampling + regulariser + dropout (main.py:206-207,260-261,110), on the captured engine and on the eager one
    for engine in ("graph", "eager"):
        f1 = cli.main(["--dataset", "cora", "--max_epochs", "1", "--runs", "1", "--batch_size", "64", "--num_samples", "8",
                       "--random_sampling", "true", "--reg_param", "0.1", "--dropout", "0.2", "--seed", "2", "--max_steps", "3",
                       "--hidden_dim", "32", "--eval_full_batch", "false", "--engine", engine])
        assert 0.0 <= f1 <= 1.0
    # GFlowNet sampler with REINFORCE (main.py:277-279) and dropout
    f1 = cli.main(["--dataset", "cora", "--max_epochs", "1", "--runs", "1", "--batch_size", "64", "--num_samples", "8",
                   "--reinforce_baseline", "true", "--dropout", "0.1", "--seed", "3", "--max_steps", "3", "--hidden_dim", "64",
                   "--eval_full_batch", "false"])
    assert 0.0 <= f1 <= 1.0


def test_edge_index_to_csr_matches_scipy_semantics():
    """Ingest: DeviceGraph.from_edge_index == sp.csr_matrix((ones, edge_index)) (main.py:134-136)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import numpy as np
    from grapes_amd.graph import DeviceGraph
    from oracle import grapes_oracle as O
    rng = np.random.default_rng(0)
    N = 300
    ei = rng.integers(0, N, (2, 4000))
    ei[:, :50] = ei[:, 50:100]                                  # duplicates collapse
    ei[1, 100:130] = ei[0, 100:130]                             # self-loops stay (the SciPy constructor keeps them)
    indptr, indices = O.build_csr(ei, N)
    g = DeviceGraph.from_edge_index(torch.from_numpy(ei), N)
    assert np.array_equal(g.rowptr.cpu().numpy(), indptr) and np.array_equal(g.col.cpu().numpy().astype(np.int64), indices)


def test_cli_trains_on_ragged_and_undersized_splits(capsys):
    """main.py:126 keeps the DataLoader's partial batches: (a) the CLI defaults (cora, batch_size 512) on a training split
    of 270 nodes are ONE ragged batch per epoch — the captured step clamps its batch size to the split; (b) batch_size 100
    leaves a 70-node tail batch per epoch, run by the eager engine on the same models / optimisers.  Both must train
    (non-zero step count and loss), and mini-batch evaluation may be called repeatedly (fresh indicator epoch per batch)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import re
    from grapes_amd import main as cli
    cli.main(["--dataset", "cora", "--max_epochs", "2", "--runs", "1", "--seed", "3", "--hidden_dim", "32"])
    out = capsys.readouterr().out
    steps = [int(m) for m in re.findall(r"(\d+) steps in", out)]
    losses = [float(m) for m in re.findall(r"loss_c=([0-9.eE+-]+)", out)]
    assert steps == [1, 1] and all(l > 0.0 for l in losses), out
    cli.main(["--dataset", "cora", "--max_epochs", "3", "--runs", "1", "--seed", "3", "--hidden_dim", "32", "--batch_size", "100",
              "--eval_frequency", "1", "--eval_full_batch", "false", "--e_cap", "16384"])
    out = capsys.readouterr().out
    steps = [int(m) for m in re.findall(r"(\d+) steps in", out)]
    assert steps == [3, 3, 3] and out.count("valid_accuracy=") == 3 and "test_accuracy=" in out, out


def test_cli_pipelined_epochs_equal_the_one_graph_ones(capsys):
    """The CLI's captured engine feeds itself (GraphedTrainer.attach_loader(epochs=True): exactly the DataLoader's full batches,
    epoch after epoch) and carries the next batch's prelude inside the current step; the ragged batch and the evaluations run on
    graph scratch of their own in between.  With --pipeline false the same loader drives the one-graph step: every epoch's
    printed losses, the validation and the test metrics are EQUAL to the last digit — across three epochs with a 14-node ragged
    batch each and an evaluation in the middle."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import re
    from grapes_amd import main as cli
    outs = []
    for pipe in ("true", "false"):
        cli.main(["--dataset", "cora", "--max_epochs", "3", "--runs", "1", "--seed", "5", "--hidden_dim", "64", "--batch_size", "64",
                  "--num_samples", "16", "--eval_frequency", "2", "--e_cap", "16384", "--pipeline", pipe])
        out = capsys.readouterr().out
        outs.append([l for l in out.splitlines() if re.search(r"loss_c=|accuracy=", l)])
    strip = lambda l: re.sub(r" in [0-9.]+s", "", l)
    assert len(outs[0]) >= 4 and [strip(l) for l in outs[0]] == [strip(l) for l in outs[1]], (outs[0], outs[1])
    steps = [int(m) for l in outs[0] for m in re.findall(r"(\d+) steps in", l)]
    assert steps == [5, 5, 5]                      # 270 training nodes: four full batches + the ragged one, every epoch


def test_cli_embed_nodes_runs_the_blogcat_style_config(capsys):
    """configs/gflownet/blogcat.txt / ogbn-proteins.txt set --embed_nodes=True (main.py:89-100,116): learned node embeddings in
    place of data.x, optimised by optimizer_c — with the file's flags (its dataset replaced by the synthetic stand-in: dataset
    files are out of scope), on the captured engine and on the eager one; the embeddings must MOVE."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import re
    from grapes_amd import main as cli
    flags = ["--batch_size=256", "--dropout=0", "--embed_nodes=True", "--eval_frequency=10", "--eval_full_batch=True",
             "--eval_on_cpu=True", "--hidden_dim=256", "--loss_coef=6414.70642460407", "--lr_gc=0.0028881609333779408",
             "--lr_gf=0.00015793805566708893", "--node_emb_dim=64", "--num_samples=256", "--sampling_hops=2",
             "--use_indicators=True"]
    for engine in ("graph", "eager"):
        f1 = cli.main(flags + ["--dataset", "cora", "--max_epochs", "2", "--runs", "1", "--seed", "4", "--e_cap", "32768",
                               "--engine", engine])
        out = capsys.readouterr().out
        assert 0.0 <= f1 <= 1.0 and "Using learned node embeddings" in out and "test_accuracy=" in out
        losses = [float(m) for m in re.findall(r"loss_c=([0-9.eE+-]+)", out)]
        assert len(losses) == 2 and all(l > 0 for l in losses)

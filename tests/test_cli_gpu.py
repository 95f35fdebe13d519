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
    # random sampling / REINFORCE / regulariser go through the eager engine (main.py:206-207,277-279,260-261)
    f1 = cli.main(["--dataset", "cora", "--max_epochs", "1", "--runs", "1", "--batch_size", "64", "--num_samples", "8",
                   "--random_sampling", "true", "--reg_param", "0.1", "--seed", "2", "--max_steps", "3", "--hidden_dim", "32",
                   "--eval_full_batch", "false"])
    assert 0.0 <= f1 <= 1.0

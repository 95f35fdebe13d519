"""Host logic of the command-line driver (grapes_amd/main.py): the reference's flags, explicit booleans and
config-file precedence (reference main.py:23-54,367-374).  No GPU involved."""
import pytest

from grapes_amd import main as cli


def test_defaults_are_the_reference_defaults():
    a = cli.parse_args([])
    assert (a.dataset, a.sampling_hops, a.num_samples, a.use_indicators) == ("cora", 2, 16, True)       # main.py:24-28
    assert (a.lr_gf, a.lr_gc, a.loss_coef, a.log_z_init, a.reg_param, a.dropout) == (1e-4, 1e-3, 1e4, 0.0, 0.0, 0.0)
    assert (a.model_type, a.hidden_dim, a.max_epochs, a.batch_size, a.eval_frequency) == ("gcn", 256, 30, 512, 5)
    assert (a.eval_on_cpu, a.eval_full_batch, a.random_sampling, a.runs, a.reinforce_baseline) == (True, True, False, 10, False)
    assert a.seed is None and a.config_file is None


def test_config_file_then_command_line_precedence(tmp_path):
    cfg = tmp_path / "products.txt"
    cfg.write_text('--batch_size 256\n--dataset "products"\n--eval_frequency 10\n--eval_full_batch true\n'
                   '--hidden_dim 256\n--log_wandb true\n--loss_coef 15227.124438334951\n--lr_gc 0.0004469352065467127\n'
                   '--lr_gf 2.5564414649576825e-05\n--max_epochs 100\n--model_type "gcn"\n--num_samples 256\n'
                   '--sampling_hops 2\n--use_indicators true\n--dropout 0.0\n--runs 10\n')
    a = cli.parse_args(["--config_file", str(cfg)])
    assert a.dataset == "products" and a.batch_size == 256 and a.num_samples == 256 and a.max_epochs == 100
    assert a.loss_coef == 15227.124438334951 and a.lr_gf == 2.5564414649576825e-05 and a.log_wandb is True
    b = cli.parse_args(["--config_file", str(cfg), "--max_epochs", "3", "--use_indicators", "false", "--runs", "1"])
    assert b.max_epochs == 3 and b.use_indicators is False and b.runs == 1 and b.batch_size == 256    # CLI wins (main.py:370-374)


def test_explicit_booleans_and_rejections():
    assert cli.parse_args(["--random_sampling", "True"]).random_sampling is True
    assert cli.parse_args(["--eval_on_cpu", "false"]).eval_on_cpu is False
    with pytest.raises(SystemExit):
        cli.parse_args(["--use_indicators", "maybe"])
    with pytest.raises(NotImplementedError):
        cli.parse_args(["--model_type", "gat"])
    e = cli.parse_args(["--embed_nodes", "true", "--node_emb_dim", "32"])       # main.py:40-41 (built in round 4)
    assert e.embed_nodes is True and e.node_emb_dim == 32

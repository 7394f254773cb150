"""The argparse surface of /root/reference/main.py:31-52 (15 flags, names, types, defaults)."""


def test_flags_and_defaults_are_the_references():
    from graphpope_amd.main import build_parser
    ns = vars(build_parser().parse_args([]))
    assert ns == {
        "dataset": "flickr", "embedding_space": "geodesic", "sampling_method": "degree_centrality",
        "num_anchor_nodes": 2, "distance_function": None, "num_workers": 6, "dropout": 0.5, "lr": 0.001,
        "num_layers": 3, "hidden_layer_size": 256, "batch_size": 1550, "epochs": 300, "seed": 42,
        "wandb_logging": False, "n_gpus": 1,
    }


def test_readme_command_line_parses_with_the_bool_quirk():
    from graphpope_amd.main import build_parser
    ns = build_parser().parse_args("--dataset pubmed --embedding_space node2vec --sampling_method stochastic "
                                   "--num_anchor_nodes 32 --distance_function euclidean --num_workers 4 --dropout 0.3 "
                                   "--lr 0.01 --num_layers 2 --hidden_layer_size 128 --batch_size 512 --epochs 3 "
                                   "--seed 7 --wandb_logging False --n_gpus 2".split())
    assert ns.dataset == "pubmed" and ns.num_anchor_nodes == 32 and ns.n_gpus == 2
    assert ns.wandb_logging is True          # type=bool: any non-empty string is truthy (main.py:49)


def test_degree_centrality_sampling_matches_networkx():
    """utils.py:38-42 reproduced on the host without NetworkX: ascending stable sort of in+out degree, last K."""
    import networkx as nx
    import numpy as np
    import torch
    from graphpope_amd import synth, utils as gp

    class D:
        pass
    ei, n = synth.rmat(8, edge_factor=4, seed=3, symmetric=False)
    ei = np.concatenate([ei, ei[:, :50], np.stack([np.arange(5), np.arange(5)])], axis=1)      # repeats + self-loops
    d = D()
    d.edge_index, d.num_nodes = torch.as_tensor(ei), n
    g = nx.DiGraph()
    g.add_nodes_from(range(n))
    g.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
    dc = nx.degree_centrality(g)
    want = list({k: v for k, v in sorted(dc.items(), key=lambda item: item[1])}.keys())[-12:]
    assert gp.sample_anchor_nodes(d, 12, "degree_centrality") == want

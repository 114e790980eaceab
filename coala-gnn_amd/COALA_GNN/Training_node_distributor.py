"""Per-step seed routing between machines ("domains").

API mirror of the reference's `Node_Distributor` (COALA-GNN-Setup/COALA_GNN/Training_node_distributor.py:5-65): same
constructor, attributes and methods, so COALA_GNN_DataLoader and user scripts use it unchanged.  Two strategies:

  "baseline"    domain k takes the k-th contiguous slice of every global batch (reference :41-47)
  "node_color"  greedy colour-affinity assignment done natively by coala_distributor_assign
                (reference :49-58 -> node_distributor_pybind.cuh:150-222)

Two things differ from the reference on purpose (SURVEY.md appendix A.1 and the domain-index note below): the per-domain
colour-count tensors have num_colors + 1 slots because colour ids run 1..num_colors with 0 = "uncoloured", and the native
distributor is told this domain's INDEX in the master list rather than the raw machine id.
"""
import torch

from COALA_GNN_Pybind import Node_distributor_pybind

__all__ = ["Node_Distributor"]

_STRATEGIES = ("baseline", "node_color")


def _pair(make):
    return [make(), make()]


class Node_Distributor(object):
    def __init__(self, comm_manager, index_tensor, batch_size, color_file: str, topk_file: str, score_file: str,
                 parsing_method="node_color"):
        if parsing_method not in _STRATEGIES:
            raise ValueError(f"Unsupported parsing method: {parsing_method} (expected one of {_STRATEGIES})")
        ids = index_tensor.detach().to("cpu")
        if ids.dtype != torch.int64:
            raise TypeError("index_tensor must hold int64 node ids (the native distributor reads them as int64_t)")
        self.comm_manager = comm_manager
        self.parsing_method = parsing_method
        self.index_tensor = ids.contiguous()   # must stay alive: the native object keeps its address
        self.index_offset = 0                  # start of the next global batch inside index_tensor

        self.batch_size = int(batch_size)
        self.domain_batch_size = self.batch_size * comm_manager.local_size
        self.global_batch_size = self.batch_size * comm_manager.global_size

        # The reference hands comm_manager.node_id to the native object, which then compares it with a position in the
        # per-master counter list (node_distributor_pybind.cuh:216).  Those agree only when machine ids are 0..n-1 in
        # master order (true for every SLURM launch of the reference); the position itself is always right.
        self.distribute_manager = Node_distributor_pybind(
            self.index_tensor.data_ptr(), comm_manager.master_process_index, self.batch_size, comm_manager.local_size,
            comm_manager.num_master_process, color_file, topk_file, score_file)
        self.num_colors = self.distribute_manager.get_num_colors()
        self.num_color_entries = self.num_colors + 1

        # double buffers: seeds of the current / next step, and colour counts being read / being gathered
        n_domains = comm_manager.num_master_process
        self.parsed_training_nodes_buffer = _pair(lambda: torch.zeros(self.domain_batch_size, dtype=torch.int64))
        self.parsed_training_nodes_buffer_header = 0
        self.cache_color_double_buffer = _pair(
            lambda: [torch.zeros(self.num_color_entries, dtype=torch.int32) for _ in range(n_domains)])
        self.cache_color_db_header = 0

    # ------------------------------------------------------------------------------------------------ colour counters
    def gather_cache_meta(self, gpu_cache_meta):
        """Sum the local GPUs' colour counters and exchange them between domain masters into the write half."""
        self.comm_manager.gather_cache_meta(gpu_cache_meta, self.cache_color_double_buffer[self.cache_color_db_header])

    # ------------------------------------------------------------------------------------------------ seeds
    def _take_baseline(self, out):
        first = self.index_offset + self.comm_manager.master_process_index * self.domain_batch_size
        out.copy_(self.index_tensor[first: first + self.domain_batch_size])

    def _take_by_color(self, out, color_buf_read_header):
        counters = self.cache_color_double_buffer[color_buf_read_header]
        self.distribute_manager.distribute_node_with_affinity(self.index_offset, out.data_ptr(),
                                                              [t.data_ptr() for t in counters])

    def parse_domain_training_nodes(self, color_buf_read_header):
        """Fill the current seed buffer with this domain's share of the next global batch and advance the cursor."""
        out = self.parsed_training_nodes_buffer[self.parsed_training_nodes_buffer_header]
        if self.index_offset + self.global_batch_size > self.index_tensor.numel():
            raise IndexError("ran past the end of the training id list (the loader stops one global batch earlier)")
        if self.parsing_method == "baseline":
            self._take_baseline(out)
        else:
            self._take_by_color(out, color_buf_read_header)
        self.index_offset += self.global_batch_size
        return out

    def reset(self):
        """Start of a new epoch."""
        self.index_offset = 0
        self.parsed_training_nodes_buffer_header = 0
        self.cache_color_db_header = 0

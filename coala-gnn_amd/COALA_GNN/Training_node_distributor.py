"""Seed-node distribution across machines ("domains").

Mirror of COALA-GNN-Setup/COALA_GNN/Training_node_distributor.py:5-65 (reference): same class, attributes and methods.
Divergence: colour-count buffers hold num_colors+1 int32 entries because colours run
1..num_colors and index 0 is "uncoloured"; the reference allocates num_colors and reads one past the end."""
import torch

from COALA_GNN_Pybind import Node_distributor_pybind

__all__ = ["Node_Distributor"]


class Node_Distributor(object):
    def __init__(self, comm_manager, index_tensor, batch_size, color_file: str, topk_file: str, score_file: str,
                 parsing_method="node_color"):
        self.index_tensor = index_tensor.to("cpu").contiguous()
        if self.index_tensor.dtype != torch.int64:
            raise TypeError("index_tensor must be int64 (node_distributor_pybind.cuh:111)")
        self.index_offset = 0
        self.parsing_method = parsing_method
        self.batch_size = batch_size
        self.comm_manager = comm_manager

        self.domain_batch_size = batch_size * comm_manager.local_size
        self.global_batch_size = batch_size * comm_manager.global_size

        # The reference passes comm_manager.node_id (:25) and compares it with an index into the per-master counter list
        # (node_distributor_pybind.cuh:216); the two agree only when machine ids are 0..n-1 in master order, so the
        # domain's index in the master list is passed instead (identical for every SLURM launch of the reference).
        self.distribute_manager = Node_distributor_pybind(
            self.index_tensor.data_ptr(), self.comm_manager.master_process_index, self.batch_size,
            comm_manager.local_size, comm_manager.num_master_process, color_file, topk_file, score_file)
        self.num_colors = self.distribute_manager.get_num_colors()
        self.num_color_entries = self.num_colors + 1

        self.parsed_training_nodes_buffer = [torch.zeros(self.domain_batch_size, dtype=torch.int64).contiguous()
                                             for _ in range(2)]
        self.parsed_training_nodes_buffer_header = 0

        self.cache_color_double_buffer = [
            [torch.zeros(self.num_color_entries, dtype=torch.int32) for _ in range(comm_manager.num_master_process)]
            for _ in range(2)]
        self.cache_color_db_header = 0

    def gather_cache_meta(self, gpu_cache_meta):
        self.comm_manager.gather_cache_meta(gpu_cache_meta, self.cache_color_double_buffer[self.cache_color_db_header])

    def parse_domain_training_nodes(self, color_buf_read_header):
        buf = self.parsed_training_nodes_buffer[self.parsed_training_nodes_buffer_header]
        if self.parsing_method == "baseline":  # :41-47 contiguous striping
            node_id = self.comm_manager.master_process_index
            lo = self.index_offset + node_id * self.domain_batch_size
            buf.copy_(self.index_tensor[lo: lo + self.domain_batch_size])
            self.index_offset += self.global_batch_size
            return buf
        elif self.parsing_method == "node_color":  # :49-58
            gather_ptr = [gt.data_ptr() for gt in self.cache_color_double_buffer[color_buf_read_header]]
            self.distribute_manager.distribute_node_with_affinity(self.index_offset, buf.data_ptr(), gather_ptr)
            self.index_offset += self.global_batch_size
            return buf
        else:
            raise ValueError(f"Unsupported parsing method: {self.parsing_method}")

    def reset(self):  # :62-65
        self.index_offset = 0
        self.parsed_training_nodes_buffer_header = 0
        self.cache_color_db_header = 0

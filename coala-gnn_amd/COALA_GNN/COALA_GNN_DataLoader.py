"""Iterator yielding (input_nodes, seeds, blocks, features) per training step.

Mirror of COALA-GNN-Setup/COALA_GNN/COALA_GNN_DataLoader.py (reference): COALA_GNN_Node_Distribution_Scheduler :8-75,
SSD_INFO :80-90, COALA_GNN_DataLoader :92-177 -- same names, arguments, cadence and double buffering.
`graph_sampler` only needs .sample(graph, seed_ids) returning a tuple whose first element is the int64 input-node
tensor (a DGL sampler object works when dgl imports; COALA_GNN.sampler.NeighborSampler is the native one)."""
import collections
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import torch

from .COALA_GNN_Manager import COALA_GNN_Manager

__all__ = ["COALA_GNN_Node_Distribution_Scheduler", "SSD_INFO", "COALA_GNN_DataLoader"]


_STREAMS = {}


def _loader_streams(device):
    """(fetch stream, sampler stream) of a device, shared by every loader of the process.  HIP maps streams onto a handful of
    hardware queues in creation order: a second loader with fresh streams can land its fetch stream on the queue of the training
    stream and lose the overlap (measured: the epoch of a loader created after another one, with streams of its own at normal
    priority, was 6 % slower)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _STREAMS:
        # (normal priority: the high level is left to the exchange's communication stream, which must not share a hardware queue
        # with the fetch stream; the epoch time is the same with either priority, 9.03-9.06 s)
        _STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _STREAMS[key]


class _NativeEvent(object):
    """The end event(s) a fetch carries on its own launches (COALA_GNN_Manager.last_done_event: one handle, or one per stream of a
    distributed fetch): same two uses as a torch event."""
    __slots__ = ("handles",)

    def __init__(self, handles):
        self.handles = tuple(handles) if isinstance(handles, (tuple, list)) else (handles,)

    def synchronize(self):
        from COALA_GNN_Pybind import event_elapsed_ms
        for h in self.handles:
            event_elapsed_ms(h, h, wait=True)


def _completed(ev):
    if isinstance(ev, _NativeEvent):
        from COALA_GNN_Pybind import event_elapsed_ms
        return all(event_elapsed_ms(h, h, wait=False) is not None for h in ev.handles)
    return ev.query()


def _wait_for(stream, ev):
    if isinstance(ev, _NativeEvent):
        from COALA_GNN_Pybind import stream_wait_event
        for h in ev.handles:
            stream_wait_event(h, int(stream.cuda_stream))
    else:
        stream.wait_event(ev)


def _device_tensors(obj):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, (list, tuple)):
        for x in obj:
            yield from _device_tensors(x)
    elif isinstance(obj, dict):
        for x in obj.values():
            yield from _device_tensors(x)
    elif hasattr(obj, "tensors"):
        for x in obj.tensors():
            yield from _device_tensors(x)


class COALA_GNN_Node_Distribution_Scheduler(object):
    """COALA_GNN_DataLoader.py:8-75.  Same two-stage pipeline (next batch's distribution and the colour-counter gather run
    behind the current step); the reference spawns a threading.Thread per step for each stage, here two persistent
    single-worker executors play those roles (a thread spawn costs ~60 us per step on this host)."""

    def __init__(self, node_distributor, ssd_gnn_manager, refresh_counter=8):
        self.node_distributor = node_distributor
        self.ssd_gnn_manager = ssd_gnn_manager
        self.metadata_reuse_counter = 0
        self.refresh_counter = refresh_counter
        self.cache_color_gathered_header = 0
        self.distribute_thread = None          # Future of the pending parse_domain_training_nodes
        self.cache_meta_gather_thread = None   # Future of the pending gather_cache_meta
        self._dist_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="coala-distribute")
        self._meta_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="coala-colour-gather")
        # num_colors + 1 entries: colours run 1..num_colors
        self.cache_meta_tensor = torch.zeros(self.node_distributor.num_colors + 1, dtype=torch.int32)

    def _distribute(self, color_header):
        """In the distribution helper, on EVERY rank: the domain master parses its domain's share of the next global batch, then the domain's
        ranks broadcast it among themselves (Shared_Tensor.py:102-103).  The reference broadcasts on the thread that drives the step
        (COALA_GNN_DataLoader.py:41-44); a gloo broadcast of a domain batch is 0.4-0.6 ms on the host, which a multi-GPU step whose fetch takes
        half a millisecond cannot hide there.  The group is used by this helper alone."""
        nd = self.node_distributor
        comm = nd.comm_manager
        if comm.is_master:
            nd.parse_domain_training_nodes(color_header)
        buf = nd.parsed_training_nodes_buffer[nd.parsed_training_nodes_buffer_header]
        comm.broadcast_training_nodes(buf)
        return buf

    def run(self, is_last: bool):  # COALA_GNN_DataLoader.py:27-75
        nd = self.node_distributor
        comm = nd.comm_manager
        if self.distribute_thread is None:  # first stage of the distribution pipeline
            self.distribute_thread = self._dist_pool.submit(self._distribute, self.cache_color_gathered_header)
        distributed_node_index = self.distribute_thread.result()
        self.distribute_thread = None
        nd.parsed_training_nodes_buffer_header = (nd.parsed_training_nodes_buffer_header + 1) % 2

        if self.metadata_reuse_counter == self.refresh_counter:
            self.metadata_reuse_counter = 0
            if self.cache_meta_gather_thread is not None:
                self.cache_meta_gather_thread.result()
                nd.cache_color_db_header = int((nd.cache_color_db_header + 1) % 2)
            cache = self.ssd_gnn_manager.COALA_GNN_Cache
            self.cache_color_gathered_header = int((nd.cache_color_db_header + 1) % 2)
            if hasattr(cache, "get_cache_data_async"):
                # The snapshot is taken at THIS point of the stream (as the reference's synchronous read, :56-58), but nobody on this
                # thread needs the numbers: the helper that gathers them across ranks waits for the copy.  A blocking read here held
                # up the thread that enqueues the next fetch for as long as the previous fetch still ran (an idle gap on the fetch
                # stream every refresh_counter steps: tools/fetch_gap_probe.py).
                cache.get_cache_data_async(self.cache_meta_tensor.numel())

                def finish_and_gather(t=self.cache_meta_tensor):
                    cache.get_cache_data_finish(t.data_ptr(), t.numel())
                    nd.gather_cache_meta(t)
                self.cache_meta_gather_thread = self._meta_pool.submit(finish_and_gather)
            else:
                cache.get_cache_data(self.cache_meta_tensor.data_ptr(), self.cache_meta_tensor.numel())
                self.cache_meta_gather_thread = self._meta_pool.submit(nd.gather_cache_meta, self.cache_meta_tensor)

        if not is_last:   # (every rank: the broadcast is part of the helper's task)
            self.distribute_thread = self._dist_pool.submit(self._distribute, self.cache_color_gathered_header)

        self.metadata_reuse_counter += 1
        local_r = comm.local_rank
        # clone: the double buffer is overwritten two steps later while the slice may still be in use
        return distributed_node_index[(local_r * nd.batch_size):((local_r + 1) * nd.batch_size)].clone()

    def drain(self):
        for f in (self.distribute_thread, self.cache_meta_gather_thread):
            if f is not None:
                f.result()
        self.distribute_thread = None
        self.cache_meta_gather_thread = None

    def __del__(self):
        try:
            self._dist_pool.shutdown(wait=False)
            self._meta_pool.shutdown(wait=False)
        except Exception:
            pass


class SSD_INFO(object):  # COALA_GNN_DataLoader.py:80-90
    def __init__(self, num_ssds, page_size, num_elems, ssd_read_offset):
        self.num_ssds = num_ssds
        self.num_elems = num_elems
        self.ssd_read_offset = ssd_read_offset
        self.page_size = page_size


class COALA_GNN_DataLoader(torch.utils.data.DataLoader):
    def __init__(self, SSD_info, node_distributor, graph, graph_sampler, batch_size, dim, fan_out, cache_size, device,
                 refresh_counter=10, cache_backend="nvshmem", sim_buf=None, shuffle=False, num_rows=None, profile=False,
                 prefetch=0, cold_partitioned=False, sync_fetch=False, counts_ahead=None):
        # like the reference, torch's DataLoader.__init__ is never called: this is a plain iterator
        # prefetch = 0: the reference's strictly serial __next__ (:149-167).  prefetch = k > 0: a producer thread runs
        # distribute -> sample -> fetch for the next k steps on its own HIP stream while the consumer trains (SURVEY f-2).
        # sync_fetch = True: the reference's fetch_feature, which returns only when the rows are in place.  False (default):
        # the fetch is only ENQUEUED on the current stream -- the training step that follows is ordered behind it by the stream,
        # and its launch overhead (about 1 ms of host time for the GraphSAGE step) overlaps the fetch instead of following it.
        # counts_ahead (native exchange + a sampler that buckets by owner; None = on whenever both are there): the count exchange of a
        # distributed fetch is issued right behind the sample, two steps before its fetch, so the fetch itself needs no host
        # synchronisation (coala_comm_counts_begin).  A fetch whose ticket has left the communicator's ring of pending count
        # exchanges falls back to the synchronous count (NativeExchange.fetch_bucketed) -- every rank makes the same calls in the same
        # order, so every rank falls back together.  Every rank of the cache group must use the same setting.
        self.counts_ahead = counts_ahead
        # one-thread pipeline: fetches kept enqueued on the fetch stream beyond the step being handed over.  1 is enough: nothing on the
        # host waits for the device, so the host runs far ahead of it anyway (2 measured level: profiles/r04_fetch_stream_packets.txt)
        self.fetch_depth = 1
        self.prefetch = int(prefetch)
        self._producer = None
        self._queue = None
        self._side_stream = None
        self._sample_stream = None
        self._stop = threading.Event()   # set by close(): the producer gives up at its next queue hand-off
        self._samples = collections.deque()   # serial mode: samples launched and not yet fetched (see _produce_one)
        self._fetched = collections.deque()   # serial mode: fetches enqueued and not yet handed over
        self._sampled = 0                # samples launched this epoch
        self.producer_times = {"schedule": 0.0, "sample": 0.0, "fetch": 0.0, "queue_full": 0.0, "gpu_backlog": 0.0}
        self.refresh_counter = refresh_counter
        self.sampler = graph_sampler
        self.batch_size = batch_size
        self.g = graph
        self.SSD_info = SSD_info
        self.cache_backend = cache_backend
        self.node_distributor = node_distributor
        self.device = device

        self.COALA_GNN_Manager = COALA_GNN_Manager(
            node_distributor=node_distributor, page_size=SSD_info.page_size, num_ssds=SSD_info.num_ssds,
            num_elems=SSD_info.num_elems, ssd_read_offset=SSD_info.ssd_read_offset, cache_size=cache_size,
            batch_size=batch_size, fan_out=fan_out, dim=dim, MPI_comm_manager=node_distributor.comm_manager, device=device,
            cache_backend=cache_backend, sim_buf=sim_buf, num_rows=num_rows, profile=profile,
            cold_partitioned=cold_partitioned, out_ring=self.prefetch + 2)  # batches alive at once: consumer + queue + producer
        self.COALA_GNN_Manager.sync_on_return = bool(sync_fetch)
        if not sync_fetch:
            # the aggregation timer samples: a timing pair on every 16th fetch, counted 16 times (two event packets less between the kernels
            # of consecutive fetches on the other 15: COALA_GNN_Manager.timing_stride)
            self.COALA_GNN_Manager.timing_stride = 16
        # The fetch kernels read tensors the sampler allocated on ITS stream.  record_stream(fetch stream) would make that safe, but the caching
        # allocator then records one event per tensor ON THE FETCH STREAM when the tensor is freed -- half a dozen barrier packets per minibatch
        # between the cold fill of one fetch and the probe of the next (30 us of idle link per step: profiles/r04_handover.txt).  Instead the
        # loader keeps every sampled batch referenced until the event behind its fetch has completed (a host-side query per step).
        self._retire = collections.deque()
        # the native sampler is stream-aware (it launches on torch's current stream); a foreign sampler keeps the caller's stream
        self._sample_on_side_stream = (not sync_fetch) and str(device).startswith("cuda") and getattr(graph_sampler, "stream_safe", False)
        # The native sampler's sample_end() / sample() return only after the host has seen the completion event of the sample's LAST kernel
        # (coala_sampler_wait): everything the fetch reads -- ids, blocks, owner counts -- is complete by then, and the fetch stream needs
        # no device-side wait for the sampler's stream (one barrier packet less in front of every probe).  The count exchange issued ahead
        # has its own event, waited for by the fetch itself.  A foreign sampler keeps the stream wait.
        self._sampler_done_on_host = bool(getattr(graph_sampler, "completes_on_host", False))
        if self.counts_ahead is None:
            self.counts_ahead = bool(getattr(graph_sampler, "bucket_by_owner", 0)) and hasattr(self.COALA_GNN_Manager.exchange, "counts_begin")
        self.counts_ahead = bool(self.counts_ahead)
        self.scheduler = COALA_GNN_Node_Distribution_Scheduler(node_distributor=self.node_distributor,
                                                               ssd_gnn_manager=self.COALA_GNN_Manager,
                                                               refresh_counter=self.refresh_counter)
        self.counter = 0
        self.index_len = len(self.node_distributor.index_tensor)
        self.total_count = int(self.index_len / self.node_distributor.global_batch_size) - 1  # :141

    def __setattr__(self, name, value):  # torch's DataLoader guards some attribute names after __init__; we never ran it
        object.__setattr__(self, name, value)

    def __iter__(self):
        return self

    def __len__(self):
        return max(self.total_count, 0)

    def _launch_sample(self):
        """scheduler -> seeds -> sample ENQUEUED on the sampler's own stream (no host wait); -> (pending sample, its event)."""
        # last step of the epoch: do not launch a distributor thread past the end of the id list (the
        # reference's is_last test compares a step counter with the id count and never fires)
        is_last_iter = self._sampled + 1 >= self.total_count
        # the colour-counter snapshot (every refresh_counter steps) is read on the fetch stream: behind the fetch enqueued last,
        # the same point of the sequence as in the reference's serial loop
        with torch.cuda.stream(self._side_stream):
            seeds = self.scheduler.run(is_last_iter)
        self._sampled += 1
        ticket = None
        with torch.cuda.stream(self._sample_stream):
            pending = self.sampler.sample_begin(self.g, seeds.to(self.device))
            xch = self.COALA_GNN_Manager.exchange
            if self.counts_ahead and pending[6] is not None and hasattr(xch, "counts_begin"):
                ticket = xch.counts_begin(pending[6][1].data_ptr())   # the owner counts the sampler just produced (device)
            ev = torch.cuda.Event()
            ev.record()
        return pending, ev, ticket

    def _enqueue_fetch(self):
        """The sample launched earlier -> its fetch ENQUEUED on the fetch stream; -> (item, event that marks its rows complete)."""
        pending, ev_s, ticket = self._samples.popleft()
        batch = self.sampler.sample_end(pending)  # counts of a sample enqueued a whole step ago: no wait in steady state
        if ticket is not None:
            batch[2][0].counts_ticket = ticket
        fs = self._side_stream
        with torch.cuda.stream(fs):
            if not self._sampler_done_on_host:
                fs.wait_event(ev_s)
            item = self.COALA_GNN_Manager.fetch_feature(batch)
            ev_f = self._done_event(fs)
        self._keep_until_fetched(batch, ev_f)   # allocated on the sampler's stream, read by the fetch kernels
        self.counter += 1
        return item, ev_f

    def _produce_one(self):
        if self._sample_on_side_stream and hasattr(self.sampler, "sample_begin"):
            # One host thread and the reference's order of calls, but nothing waits on the host: the sampler runs on its own
            # stream two steps ahead, the fetch on a second stream one step ahead.  __next__(t) hands over the rows whose fetch
            # was enqueued during __next__(t-1) -- after it has enqueued fetch t+1 and sample t+2 -- so the PCIe-bound fill of
            # the next minibatch runs beside the training kernels of this one, and the sampler's counts have long arrived when
            # they are asked for.  (The scheduler's calls move ahead by two steps; they touch nothing a training step does, and
            # the colour snapshot stays ordered behind the same fetch as in the serial sequence.)
            if self._side_stream is None:
                self._side_stream, self._sample_stream = _loader_streams(self.device)
            if not self._fetched:  # first call of an epoch: fill the pipeline
                self._pump()
            item, ev = self._fetched.popleft()
            self._pump()
            cur = torch.cuda.current_stream()
            _wait_for(cur, ev)  # the consumer's stream sees the finished rows
            for t in _device_tensors(item):
                t.record_stream(cur)  # allocated on the side streams, used by the training step on this one
            return item
        is_last_iter = self.counter + 1 >= self.total_count
        seeds = self.scheduler.run(is_last_iter)
        batch = self.sampler.sample(self.g, seeds.to(self.device))
        self.counter += 1
        return self.COALA_GNN_Manager.fetch_feature(batch)

    def _done_event(self, stream):
        """The event that marks the rows of the fetch just enqueued complete: the one its last kernel carries, when the manager has
        one (no packet of its own in the fetch stream's queue), else an event recorded behind it."""
        nat = getattr(self.COALA_GNN_Manager, "last_done_event", None)
        if isinstance(nat, torch.cuda.Event):
            return nat                       # (the manager had to record one itself: reuse it)
        if nat:
            return _NativeEvent(nat)
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def _keep_until_fetched(self, batch, ev):
        """Hold the sampled batch until the event behind its fetch has completed; let go of the ones whose fetch has (in order)."""
        self._retire.append((ev, batch))
        while self._retire and _completed(self._retire[0][0]):
            self._retire.popleft()

    def _retire_all(self):
        """End of an epoch / close(): nothing of this loader may still be read by a queued fetch when its tensors go back to the allocator."""
        pending = getattr(self, "_retire", None)
        if pending:
            pending[-1][0].synchronize()
            pending.clear()

    def _pump(self):
        """Keep fetch_depth fetches enqueued and one sample launched beyond the last of them (same order of calls as ever: fetch t+1,
        then sample t+2)."""
        while len(self._fetched) < self.fetch_depth:
            if not self._samples:
                if self._sampled >= self.total_count:
                    break
                self._samples.append(self._launch_sample())
            self._fetched.append(self._enqueue_fetch())
        if not self._samples and self._sampled < self.total_count:
            self._samples.append(self._launch_sample())

    def _end_of_epoch(self):
        self._samples.clear()
        self._fetched.clear()
        self._retire_all()
        self._sampled = 0
        self.scheduler.drain()  # the reference resets while a distributor thread may still run
        self.node_distributor.reset()
        self.counter = 0

    def _producer_loop(self):
        """Two stages on two HIP streams: the sampler of step t+1 runs (and its count read-back blocks this thread) while the
        fetch of step t -- enqueued without a host wait -- is still pulling rows over PCIe."""
        try:
            torch.cuda.set_device(self.device)
            mgr = self.COALA_GNN_Manager
            keep_sync = mgr.sync_on_return
            mgr.sync_on_return = False  # the consumer waits on an event instead

            T = self.producer_times  # host seconds per stage, cumulative (diagnostics; tools/prefetch_probe.py)
            clock = time.perf_counter

            def sample_next():
                t0 = clock()
                is_last_iter = self.counter + 1 >= self.total_count
                # the colour-counter snapshot (every refresh_counter steps) is read on the fetch stream: ordered after the
                # fetch enqueued last -- the same point of the sequence as in the serial loader -- and not behind the
                # consumer's training kernels on the default stream
                with torch.cuda.stream(self._side_stream):
                    seeds = self.scheduler.run(is_last_iter)
                t1 = clock()
                with torch.cuda.stream(self._sample_stream):
                    batch = self.sampler.sample(self.g, seeds.to(self.device))
                    ev = torch.cuda.Event()
                    ev.record(self._sample_stream)
                self.counter += 1
                T["schedule"] += t1 - t0
                T["sample"] += clock() - t1
                return batch, ev

            nxt = sample_next() if self.counter < self.total_count else None
            in_flight = collections.deque()
            while nxt is not None:
                batch, ev_s = nxt
                # flow control: at most `prefetch` fetches queued on the GPU.  Without it both host threads run many steps
                # ahead of the device (nothing below waits on the fetch stream), every step in flight pins ~150 MB of rows,
                # and the first call that does wait pays for the whole backlog.
                if len(in_flight) >= self.prefetch:
                    t0 = clock()
                    in_flight.popleft().synchronize()
                    T["gpu_backlog"] += clock() - t0
                t0 = clock()
                with torch.cuda.stream(self._side_stream):
                    if not self._sampler_done_on_host:
                        self._side_stream.wait_event(ev_s)
                    item = mgr.fetch_feature(batch)
                    ev_f = self._done_event(self._side_stream)
                self._keep_until_fetched(batch, ev_f)   # allocated on the sampler's stream, read by the fetch kernels
                in_flight.append(ev_f)
                T["fetch"] += clock() - t0
                nxt = sample_next() if self.counter < self.total_count else None  # overlaps the fetch just enqueued
                t0 = clock()
                if not self._hand_over((item, ev_f)):
                    break  # the loader is being closed mid-epoch
                T["queue_full"] += clock() - t0
            mgr.sync_on_return = keep_sync
            self._hand_over(None)
        except BaseException as e:  # surface producer failures in the consumer
            self._hand_over(e)

    def _hand_over(self, obj):
        """queue.put that a close() can interrupt (a consumer that left the loop early never empties the queue)."""
        while not self._stop.is_set():
            try:
                self._queue.put(obj, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def close(self):
        """Stop a prefetching producer (if any) and the scheduler's helpers; safe to call twice, also mid-epoch."""
        self._stop.set()
        if self._producer is not None:
            try:
                while True:  # let a producer blocked on a full queue see the flag
                    self._queue.get_nowait()
            except queue.Empty:
                pass
            self._producer.join(timeout=10)
            self._producer = None
        self._retire_all()
        self.scheduler.drain()

    def __next__(self):  # COALA_GNN_DataLoader.py:149-167
        if self.prefetch <= 0:
            if self.counter >= self.total_count and not self._fetched:
                self._end_of_epoch()
                raise StopIteration
            return self._produce_one()
        if self._producer is None:
            if self.total_count <= 0:
                raise StopIteration
            self._queue = queue.Queue(maxsize=self.prefetch)
            if self._side_stream is None:
                # (stream priorities were tried: a high-priority sampler stream shortens the sampler's host wait from 1.5 to
                # 0.3 ms under a training load but leaves the epoch time unchanged -- the fetch, not the sampler, is the limit)
                self._side_stream, self._sample_stream = _loader_streams(self.device)
            self._stop.clear()
            self._producer = threading.Thread(target=self._producer_loop, daemon=True)
            self._producer.start()
        got = self._queue.get()
        if got is None or isinstance(got, BaseException):
            self._producer.join()
            self._producer = None
            self._end_of_epoch()
            if got is None:
                raise StopIteration
            raise got
        item, ev = got
        cur = torch.cuda.current_stream()
        _wait_for(cur, ev)  # the consumer's stream sees the finished rows
        # Everything in the item was allocated on the producer's stream: tell the caching allocator that the consumer's
        # stream uses it too, or the memory can be handed to the producer's next step while consumer kernels still read it
        # (seen as garbage neighbour indices -> device-side index assert).
        for t in _device_tensors(item):
            t.record_stream(cur)
        return item

    def print_stats(self):  # :170-174
        self.COALA_GNN_Manager.print_stats()
        agg_time = self.COALA_GNN_Manager.get_aggregate_time()
        print(f"Aggregation time: {agg_time}")
        self.COALA_GNN_Manager.aggregation_timer = 0.0

    def __del__(self):
        try:
            self.close()
            del self.COALA_GNN_Manager
        except Exception:
            pass

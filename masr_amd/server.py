"""Batched serving front-end: one ``EngineWorker`` per GPU, a router in front of them.

The reference server owns ONE MASRPredictor and runs every request inline in the asyncio handler: one utterance per
``predict`` call and at most one websocket stream at a time (``predictor.running``; infer_server.py:42-46,48-71,103-156).
The MI355X engine only reaches its throughput with batches, so this front-end keeps the reference's wire protocol --

    POST /recognition             multipart field ``audio``  -> {"code": 0, "msg": "success", "result": text, "score": s}
    POST /recognition_long_audio  multipart field ``audio``  -> same, through ``predict_long``
    websocket /                   binary PCM chunks, the last one ending in b'end' -> {"code": 0, "result": text} per chunk
    failures                      {"error": 1, "msg": "audio read fail!"} / {"code": 2, "msg": "recognition fail!"}

-- and puts an ``EngineWorker`` between the handlers and the predictor:

* offline requests are collected for up to ``max_wait_ms`` (or until ``max_batch`` are waiting) and recognised in ONE
  ``predict_batch`` call (dynamic batching);
* every websocket is its own ``StreamPool`` session; chunks that arrive within a tick are fed together and advanced by ONE
  ``StreamPool.step()`` (one ragged fbank launch, lock-step ``masr_encode_chunk`` calls);
* the worker is the only thread that touches the engine (the C-ABI handle is not thread-safe, include/masr_hip.h).

Several GPUs (BASELINE configs[4]: 128 streams sticky over the 8 GPUs of a node): ``create_app(predictors=[...])`` with one
predictor per GPU builds one worker thread per engine in ONE process (every C-ABI entry point selects its engine's device;
the worker thread makes it torch's current device too) behind a ``WorkerRouter``: a websocket session lives on the worker it
was opened on for its whole life (its caches never move), sessions go round-robin over the workers; an offline request goes
to the worker with the least queued work.  Nothing is exchanged between the GPUs: the path's only collective (the all-gather
of hypotheses, parallel.py) belongs to batch jobs, not to a request/response server.

No static pages, templates or upload archive: the web UI is outside the hot path.  ``uvicorn`` needs the ``websockets`` or
``wsproto`` package to serve the websocket route; the in-process test client does not.
"""
import asyncio
import logging
import os
import threading
import time
from concurrent.futures import Future
from email.parser import BytesParser
from email.policy import HTTP

logger = logging.getLogger(__name__)


class EngineWorker(object):
    """Single thread that owns the predictor (and its StreamPool): dynamic batching of offline requests, lock-step stepping
    of stream sessions.  All public methods are thread-safe and return ``concurrent.futures.Future`` objects."""

    def __init__(self, predictor, pool=None, max_batch=32, max_wait_ms=10.0):
        self.predictor = predictor
        self.pool = pool
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1000.0
        self._cv = threading.Condition()
        self._offline = []          # (arrival time, audio, future)
        self._calls = []            # (fn, args, future): run on the worker thread as they are (stream open/close, predict_long)
        self._feeds = []            # (handle, pcm bytes, is_end, future)
        self._stop = False
        self.stats = {'batches': 0, 'utterances': 0, 'steps': 0, 'chunks': 0}
        self._busy = 0              # requests taken off the queues and not finished yet
        self._thread = threading.Thread(target=self._run, name='masr-engine-worker', daemon=True)
        self._thread.start()

    # ---- client side --------------------------------------------------------------------------------------------------
    def recognize(self, audio):
        """one utterance (path / wav bytes / ndarray) -> Future of {'text', 'score'}; batched with its neighbours"""
        fut = Future()
        with self._cv:
            self._offline.append((time.monotonic(), audio, fut))
            self._cv.notify()
        return fut

    def call(self, fn, *args, **kwargs):
        """run ``fn(*args, **kwargs)`` on the worker thread (anything else that touches the engine)"""
        fut = Future()
        with self._cv:
            self._calls.append((fn, args, kwargs, fut))
            self._cv.notify()
        return fut

    def recognize_long(self, audio, **kwargs):
        return self.call(self.predictor.predict_long, audio, **kwargs)

    def stream_open(self):
        return self.call(self.pool.open)

    def stream_close(self, handle):
        return self.call(self.pool.close, handle)

    def stream_feed(self, handle, pcm_bytes, is_end=False):
        """queue one chunk of a session -> Future of {'text', 'score'} or None (not enough audio for a window yet)"""
        fut = Future()
        with self._cv:
            self._feeds.append((handle, pcm_bytes, bool(is_end), fut))
            self._cv.notify()
        return fut

    def load(self):
        """queued + running requests (what the router balances offline requests on)"""
        with self._cv:
            return len(self._offline) + len(self._feeds) + len(self._calls) + self._busy

    def shutdown(self):
        with self._cv:
            self._stop = True
            self._cv.notify()
        self._thread.join()

    # ---- worker side --------------------------------------------------------------------------------------------------
    def _take(self, block=True):
        """block until there is something to do (``block=False``: look once); returns (calls, offline batch, feeds)"""
        with self._cv:
            while True:
                if self._stop:
                    return None
                now = time.monotonic()
                batch_ready = self._offline and (len(self._offline) >= self.max_batch or
                                                 now - self._offline[0][0] >= self.max_wait)
                if self._calls or self._feeds or batch_ready:
                    calls, self._calls = self._calls, []
                    feeds, self._feeds = self._feeds, []
                    batch = []
                    if batch_ready:
                        batch, self._offline = self._offline[:self.max_batch], self._offline[self.max_batch:]
                    self._busy = len(calls) + len(batch) + len(feeds)
                    return calls, batch, feeds
                if not block:
                    return [], [], []
                timeout = None
                if self._offline:
                    timeout = max(0.0, self.max_wait - (now - self._offline[0][0]))
                self._cv.wait(timeout)

    def _batch_ready(self):
        with self._cv:
            return bool(self._offline) and (len(self._offline) >= self.max_batch or
                                            time.monotonic() - self._offline[0][0] >= self.max_wait)

    def _enter_device(self):
        """this thread works for ONE engine: make its GPU torch's current device here (allocations, current stream)"""
        try:
            dev = self.predictor.predictor.engine.device
            import torch
            if dev.type == 'cuda':
                torch.cuda.set_device(dev)
        except AttributeError:
            pass                      # stand-in predictors of the CPU tests have no engine

    def _run(self):
        self._enter_device()
        # offline batches are LAUNCHED (MASRPredictor.predict_batch_deferred) and collected afterwards: while one is on the device
        # the next one that is ready is staged, uploaded and launched (on the engine's other lane) before the first is waited for --
        # at most two in flight, and nothing is held back when no further batch is ready
        inflight = []
        while True:
            work = self._take(block=not inflight)
            if work is None:
                for entry in inflight:
                    self._finish_batch(entry)
                return
            calls, batch, feeds = work
            for fn, args, kwargs, fut in calls:
                self._resolve(fut, fn, *args, **kwargs)
            if feeds:
                self._step_streams(feeds)
            if batch:
                if len(inflight) >= 2:
                    self._finish_batch(inflight.pop(0))
                entry = self._start_batch(batch)
                if entry is not None:
                    inflight.append(entry)
                    if self._batch_ready():
                        with self._cv:
                            self._busy = sum(len(e[0]) for e in inflight)
                        continue
            while inflight:
                self._finish_batch(inflight.pop(0))
            with self._cv:
                self._busy = 0

    @staticmethod
    def _resolve(fut, fn, *args, **kwargs):
        if not fut.set_running_or_notify_cancel():
            return
        try:
            fut.set_result(fn(*args, **kwargs))
        except BaseException as e:          # the client gets the exception, the worker lives on
            fut.set_exception(e)

    def _start_batch(self, batch):
        """launch one offline batch; -> (live requests, function that returns their results) or None when nothing is left to wait
        for (cancelled requests, a predictor without deferred passes, a batch that failed at launch and was redone one by one)"""
        live = [(a, f) for _, a, f in batch if f.set_running_or_notify_cancel()]
        if not live:
            return None
        self.stats['batches'] += 1
        self.stats['utterances'] += len(live)
        # (MASR_SERVE_PIPELINE=0, A/B: every batch recognised and waited for before the next one is looked at, rounds 3-5)
        deferred = getattr(self.predictor, 'predict_batch_deferred', None) if os.environ.get('MASR_SERVE_PIPELINE', '1') == '1' else None
        try:
            if deferred is not None and len(live) > 1:
                return live, deferred([a for a, _ in live])
            if len(live) == 1:
                results = [self.predictor.predict(audio_data=live[0][0])]
            else:
                results = self.predictor.predict_batch([a for a, _ in live])
            for (_, fut), res in zip(live, results):
                fut.set_result(res)
        except BaseException:
            self._one_by_one(live)
        return None

    def _finish_batch(self, entry):
        live, fetch = entry
        try:
            for (_, fut), res in zip(live, fetch()):
                fut.set_result(res)
        except BaseException:
            self._one_by_one(live)

    def _one_by_one(self, live):
        # one unreadable upload must not fail its neighbours: fall back to one call per request
        for audio, fut in live:
            if fut.done():
                continue
            try:
                fut.set_result(self.predictor.predict(audio_data=audio))
            except BaseException as e:
                fut.set_exception(e)

    def _step_streams(self, feeds):
        # a session may have queued several chunks since the last tick: they are stepped in arrival order, one chunk per
        # session per step, so that every chunk gets the partial result predict_stream would have returned for it
        while feeds:
            now, later, seen = [], [], set()
            for item in feeds:
                (later if item[0] in seen else now).append(item)
                seen.add(item[0])
            live = []
            for handle, data, is_end, fut in now:
                if not fut.set_running_or_notify_cancel():
                    continue
                try:
                    self.pool.feed(handle, data, is_end=is_end)
                    live.append((handle, fut))
                except BaseException as e:
                    fut.set_exception(e)
            if live:
                self.stats['steps'] += 1
                self.stats['chunks'] += len(live)
                try:
                    out = self.pool.step()
                    errors = getattr(self.pool, 'errors', {})
                    for handle, fut in live:
                        if handle in errors:        # the step left this session out (its stream is full): only ITS request fails
                            fut.set_exception(Exception(errors[handle]))
                        else:
                            fut.set_result(out.get(handle))
                except BaseException as e:
                    for _, fut in live:
                        fut.set_exception(e)
            feeds = later


class WorkerRouter(object):
    """N EngineWorkers (one per GPU) behind the interface of one.  Stream sessions are sticky: handle = local handle * N +
    worker index, opened round-robin; offline requests go to the least loaded worker (ties: lowest index)."""

    def __init__(self, workers):
        self.workers = list(workers)
        if not self.workers:
            raise ValueError('WorkerRouter needs at least one worker')
        self._lock = threading.Lock()
        self._next = 0

    @property
    def n(self):
        return len(self.workers)

    @property
    def stats(self):
        out = {}
        for w in self.workers:
            for k, v in w.stats.items():
                out[k] = out.get(k, 0) + v
        out['per_worker'] = [dict(w.stats) for w in self.workers]
        return out

    def _pick(self):
        loads = [w.load() for w in self.workers]
        return self.workers[loads.index(min(loads))]

    def recognize(self, audio):
        return self._pick().recognize(audio)

    def recognize_long(self, audio, **kwargs):
        return self._pick().recognize_long(audio, **kwargs)

    def call(self, fn, *args, **kwargs):
        return self.workers[0].call(fn, *args, **kwargs)

    def owner(self, handle):
        return handle % self.n

    def stream_open(self):
        with self._lock:
            i = self._next % self.n
            self._next += 1
        inner, out = self.workers[i].stream_open(), Future()

        def done(f):
            try:
                out.set_result(f.result() * self.n + i)
            except BaseException as e:
                out.set_exception(e)
        inner.add_done_callback(done)
        return out

    def stream_feed(self, handle, pcm_bytes, is_end=False):
        return self.workers[handle % self.n].stream_feed(handle // self.n, pcm_bytes, is_end)

    def stream_close(self, handle):
        return self.workers[handle % self.n].stream_close(handle // self.n)

    def shutdown(self):
        for w in self.workers:
            w.shutdown()


def _multipart_file(content_type, body, field='audio'):
    """the bytes of multipart form field ``field`` (python-multipart is not a dependency); raw bodies pass through"""
    if not content_type or 'multipart/form-data' not in content_type:
        return body
    msg = BytesParser(policy=HTTP).parsebytes(b'Content-Type: ' + content_type.encode() + b'\r\n\r\n' + body)
    for part in msg.iter_parts():
        if part.get_param('name', header='content-disposition') == field:
            return part.get_payload(decode=True)
    raise ValueError(f'multipart field {field!r} missing')


def create_app(predictor=None, max_batch=32, max_wait_ms=10.0, max_frames_out=0, pool=None, predictors=None, pools=None):
    """FastAPI application speaking the reference server's protocol on top of an EngineWorker -- or, with
    ``predictors=[one MASRPredictor per GPU]``, on a ``WorkerRouter`` over one worker per GPU (sticky websocket sessions,
    least-loaded offline requests).  ``app.state.worker`` is the worker / router (``.stats`` counts batches / utterances /
    steps / chunks).  ``pool`` / ``pools``: the StreamPool(s) of the websocket sessions (default: one is built per streaming
    ctc_greedy predictor)."""
    from fastapi import FastAPI, Request, WebSocket
    from starlette.websockets import WebSocketDisconnect

    if predictors is None:
        if predictor is None:
            raise ValueError('create_app needs a predictor (or predictors=[...])')
        predictors, pools = [predictor], [pool]
    elif pools is None:
        pools = [None] * len(predictors)
    workers = []
    for pr, pl in zip(predictors, pools):
        cfg = pr.configs
        if pl is None and cfg.streaming and cfg.decoder == 'ctc_greedy':
            from masr_amd.serving import StreamPool
            pl = StreamPool(pr, max_frames_out=max_frames_out)
        workers.append(EngineWorker(pr, pl, max_batch=max_batch, max_wait_ms=max_wait_ms))
    worker = workers[0] if len(workers) == 1 else WorkerRouter(workers)
    pool = workers[0].pool if all(w.pool is not None for w in workers) else None
    app = FastAPI(title='MASR')
    app.state.worker = worker

    async def _upload(request):
        return _multipart_file(request.headers.get('content-type'), await request.body())

    @app.post('/recognition')
    async def recognition(request: Request):
        try:
            res = await asyncio.wrap_future(worker.recognize(await _upload(request)))
            return {'code': 0, 'msg': 'success', 'result': res['text'], 'score': round(res['score'], 3)}
        except Exception:
            return {'error': 1, 'msg': 'audio read fail!'}

    @app.post('/recognition_long_audio')
    async def recognition_long_audio(request: Request):
        try:
            res = await asyncio.wrap_future(worker.recognize_long(await _upload(request)))
            return {'code': 0, 'msg': 'success', 'result': res['text'], 'score': res['score']}
        except FileNotFoundError as exc:
            # the Silero weights (silero_vad.onnx) are third-party and not shipped: say so instead of blaming the upload
            logger.error(f'/recognition_long_audio: {exc}')
            return {'error': 2, 'msg': f'VAD model not found: {exc}'}
        except Exception as exc:                                        # noqa: BLE001  (same reply as the reference server)
            logger.exception(f'/recognition_long_audio failed: {exc}')
            return {'error': 1, 'msg': 'audio read fail!'}

    @app.websocket('/')
    async def websocket_endpoint(websocket: WebSocket):
        await websocket.accept()
        if pool is None:
            await websocket.send_json({'code': 1, 'msg': 'recognition fail, no resource!'})
            await websocket.close()
            return
        handle = await asyncio.wrap_future(worker.stream_open())
        text = ''
        try:
            while True:
                data = await websocket.receive_bytes()
                if len(data) == 0:
                    continue
                is_end = data[-3:] == b'end'
                if is_end:
                    data = data[:-3]
                try:
                    res = await asyncio.wrap_future(worker.stream_feed(handle, data, is_end))
                    if res is not None:
                        text = res['text']
                    await websocket.send_json({'code': 0, 'result': text})
                except WebSocketDisconnect:
                    raise
                except Exception:
                    await websocket.send_json({'code': 2, 'msg': 'recognition fail!'})
                if is_end:
                    await websocket.close()
                    break
        except WebSocketDisconnect:
            pass
        finally:
            await asyncio.wrap_future(worker.stream_close(handle))

    @app.on_event('shutdown')
    def _shutdown():
        worker.shutdown()

    return app

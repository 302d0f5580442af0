import sys, time, threading, traceback, runpy
sys.path.insert(0, '.')
main_id = threading.get_ident()
stop = False
def watch():
    last_top, since = None, time.perf_counter()
    dumped = 0
    while not stop:
        time.sleep(0.002)
        fr = sys._current_frames().get(main_id)
        if fr is None: continue
        top = (fr.f_code.co_filename, fr.f_lineno)
        now = time.perf_counter()
        if top != last_top:
            last_top, since = top, now
        elif now - since > 0.015 and "bench.py" in "".join(f.filename for f in traceback.extract_stack(fr)) and dumped < 6:
            st = traceback.extract_stack(fr)
            if any("fuse_and_decode_async" in f.name for f in st):
                print("STALL %.1f ms at:" % (1e3 * (now - since)), " <- ".join(f"{f.name}:{f.lineno}" for f in st[-5:]), file=sys.stderr)
                dumped += 1
                since = now
threading.Thread(target=watch, daemon=True).start()
sys.argv = ["bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-alt-mode"]
runpy.run_path("bench.py", run_name="__main__")
stop = True

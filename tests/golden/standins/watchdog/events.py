class FileSystemEventHandler:
    pass

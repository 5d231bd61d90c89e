from .components import main

main()
